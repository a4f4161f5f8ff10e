#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b8
timeout 600 python3 scratch/efused_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b8/efused_check.txt
timeout 900 python3 -m pytest tests/test_gpu_energy.py -m gpu -x -q > gpurun_out/b8/tests_energy.txt 2>&1
cat gpurun_out/b8/efused_check.txt; tail -12 gpurun_out/b8/tests_energy.txt
