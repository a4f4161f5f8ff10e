#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b17; mkdir -p $O
timeout 900 python3 scratch/r04_time3.py default mfma_ilp mfma_memcl 2>/dev/null | tee $O/time3.txt
timeout 600 python3 -m pytest tests/test_gpu_grad.py -q -x -k "large_batch_training" > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -3 $O/tests.txt
