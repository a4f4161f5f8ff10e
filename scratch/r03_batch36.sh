#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b36
timeout 1500 python3 -m pytest tests/test_gpu_inverse.py -m gpu -q -k "more_than_32 or staged" > gpurun_out/b36/tests.txt 2>&1
tail -30 gpurun_out/b36/tests.txt
