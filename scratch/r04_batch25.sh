#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b25; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_energy.py tests/test_gpu_inverse.py -q > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -4 $O/tests.txt
