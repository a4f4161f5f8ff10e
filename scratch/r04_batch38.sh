#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b38; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_grad.py -q -x -k "large_batch_training" > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -3 $O/tests.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 scratch/r04_train_trace.py > $O/trace.log 2>&1
python3 scratch/r04_trace_gaps.py $O/trace | tee $O/train_step_timeline.txt
rm -rf $O/trace
