#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b9
( time timeout 900 python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > gpurun_out/b9/smoke.txt 2>&1
( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/b9/bench_line.txt 2>&1
bash scratch/prof_r03.sh > gpurun_out/b9/prof.txt 2>&1
grep -v amdgpu.ids gpurun_out/b9/smoke.txt | tail -12; grep -E "^real|^\{" gpurun_out/b9/bench_line.txt | cut -c1-200; tail -30 gpurun_out/b9/prof.txt | cut -c1-220
