#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b13; mkdir -p $O
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a $O/lines.txt; done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_regime.json 2> $O/bench.err
python3 -c "import sys,json; d=json.loads(open('$O/bench_line_driver_regime.json').read().strip().splitlines()[-1]); print('full', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a $O/lines.txt
