#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b27; mkdir -p $O
REPS=40 timeout 600 python3 scratch/r04_repro_diag.py 2>/dev/null | tee $O/repro.txt
timeout 1500 python3 -m pytest tests/test_gpu_grad.py -q > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -4 $O/tests.txt
timeout 300 python3 scratch/r04_grad33_time.py 2>/dev/null | grep matrix | tee $O/time.txt
