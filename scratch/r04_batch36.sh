#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b36; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_inverse.py -q -x > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -6 $O/tests.txt
timeout 300 python3 scratch/r04_sample_time.py 2>/dev/null | tee $O/sample_new.txt
WF_SAMPLE_ONE_LANE=1 timeout 300 python3 scratch/r04_sample_time.py 2>/dev/null | tee $O/sample_onelane.txt
timeout 300 python3 scratch/r04_sample_time.py 2>/dev/null | tee -a $O/sample_new.txt
