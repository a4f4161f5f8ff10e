#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b34
timeout 900 python3 scratch/tsample_check.py 2>&1 | grep -v amdgpu.ids | grep -E "exact=True|time|latent1|x1" | head -6 | tee gpurun_out/b34/tsample_check.txt
timeout 1500 python3 -m pytest tests/test_gpu_inverse.py -m gpu -x -q > gpurun_out/b34/tests.txt 2>&1
tail -3 gpurun_out/b34/tests.txt
timeout 900 python3 scratch/tsample_ks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b34/ks.txt
bash scratch/r03_batch29.sh
