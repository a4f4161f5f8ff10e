#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b18; mkdir -p $O
ROUNDS=4 timeout 900 python3 scratch/time_ab.py mfma_memcl default mfma_ilp default mfma_memcl 2>/dev/null | tee $O/time_ab.txt
timeout 600 python3 -m pytest tests/test_gpu_grad.py -q -x -k "large_batch_training" > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -3 $O/tests.txt
