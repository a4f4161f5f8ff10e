#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b6; mkdir -p $O
timeout 1200 python3 scratch/r04_d8_time.py > $O/d8_time.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "c4 or reproducible or strict or other_dims or staged" > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt
cat $O/d8_time.txt; tail -3 $O/tests.txt
