#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b6
timeout 300 scratch/ubench2/pk_mfma_roles.bin > gpurun_out/b6/pk_mfma_roles.txt 2>&1
timeout 600 python3 scratch/nsc_err_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b6/nsc_err.txt
timeout 600 python3 -m pytest tests/test_gpu_rqs.py -m gpu -x -q > gpurun_out/b6/tests_rqs.txt 2>&1
cat gpurun_out/b6/pk_mfma_roles.txt gpurun_out/b6/nsc_err.txt; tail -15 gpurun_out/b6/tests_rqs.txt
