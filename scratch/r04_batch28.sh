#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b28; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_grad.py -q -x -k "two_row_blocks or tile_path" > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -4 $O/tests.txt
timeout 600 python3 bench.py --workload vqmc --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_line_vqmc.json 2>$O/bench_vqmc.err; tail -c 1500 $O/bench_line_vqmc.json
