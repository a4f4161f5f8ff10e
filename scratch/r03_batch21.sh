#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b21
timeout 1500 python3 -m pytest tests/test_gpu_grad.py -m gpu -x -q -k "large_batch_training or training_reduces or two_rank or graph_capturable" > gpurun_out/b21/tests.txt 2>&1
tail -15 gpurun_out/b21/tests.txt
