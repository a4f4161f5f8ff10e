#!/bin/bash
# the GPU suite three times in a row on one box (intermittent failures show as a difference between the runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
for i in 1 2 3; do
  timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/soak/tests_$i.txt 2>&1
  tail -2 gpurun_out/soak/tests_$i.txt
done
