#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b4
timeout 600 python3 scratch/efused_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b4/efused_check.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/b4/tests.txt 2>&1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b4/bench_line.txt 2>&1
bash scratch/prof_r03.sh > gpurun_out/b4/prof.txt 2>&1
cat gpurun_out/b4/efused_check.txt; tail -4 gpurun_out/b4/tests.txt; tail -1 gpurun_out/b4/bench_line.txt | cut -c1-3000; cat gpurun_out/b4/prof.txt
