#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b42; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_energy.py -q > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -4 $O/tests.txt
for i in 1 2; do timeout 300 python3 scratch/r04_grad33_time.py 2>/dev/null | grep matrix; done | tee $O/time.txt
REPS=500 timeout 600 python3 scratch/r04_soak.py 2>/dev/null | tee $O/soak.txt
