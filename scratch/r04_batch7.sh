#!/bin/bash
# two row blocks per dimension through the matrix-core gradient path + the shared, ticketed accumulators: tests, then times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b7; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_grad.py -q -x -k "matrix_cores or tile_path or large_batch_training or captured_large" > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt
tail -15 $O/tests.txt
timeout 600 python3 scratch/r04_grad33_time.py > $O/grad_time.txt 2>&1
cat $O/grad_time.txt
