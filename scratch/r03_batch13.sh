#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b13
timeout 1500 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_energy.py -m gpu -x -q > gpurun_out/b13/tests.txt 2>&1
tail -25 gpurun_out/b13/tests.txt
