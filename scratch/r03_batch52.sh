#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b52
timeout 2400 python3 -m pytest tests/test_gpu_grad.py -m gpu -x -q > gpurun_out/b52/tests.txt 2>&1
tail -6 gpurun_out/b52/tests.txt
