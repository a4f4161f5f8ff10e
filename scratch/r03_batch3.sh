#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b3
timeout 600 python3 scratch/efused_check.py > gpurun_out/b3/efused_check.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_energy.py tests/test_gpu_grad.py -m gpu -x -q > gpurun_out/b3/tests_energy.txt 2>&1
cat gpurun_out/b3/efused_check.txt | grep -v amdgpu.ids; tail -5 gpurun_out/b3/tests_energy.txt
