#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b7
timeout 900 python3 -m pytest tests/test_gpu_grad.py -m gpu -x -q -k "nonzero_boundary_value or abi_error_paths_of_the_gradient" > gpurun_out/b7/tests_bias.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/b7/tests.txt 2>&1
tail -30 gpurun_out/b7/tests_bias.txt; tail -8 gpurun_out/b7/tests.txt
