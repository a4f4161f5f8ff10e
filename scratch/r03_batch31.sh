#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b31
timeout 1500 python3 -m pytest tests/test_gpu_inverse.py -m gpu -x -q > gpurun_out/b31/tests.txt 2>&1
tail -5 gpurun_out/b31/tests.txt
