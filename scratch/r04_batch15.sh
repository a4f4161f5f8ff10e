#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b15; mkdir -p $O; rm -f $O/lines.txt
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
echo "start: $(thr)" | tee -a $O/lines.txt
for i in 1 2 3 4 5 6 7 8; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a $O/lines.txt; echo "   $(thr)" | tee -a $O/lines.txt; done
timeout 1500 python3 -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -3 $O/tests.txt; echo "after tests: $(thr)" | tee -a $O/lines.txt
