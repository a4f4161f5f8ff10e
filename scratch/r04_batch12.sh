#!/bin/bash
# full GPU suite + smoke + driver-regime bench line on the current tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b12; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt
tail -6 $O/tests.txt
timeout 600 python3 __graft_entry__.py --smoke > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -8 $O/smoke.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_regime.json 2> $O/bench.err; tail -c 3000 $O/bench_line_driver_regime.json
