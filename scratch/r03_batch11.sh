#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b11
timeout 600 python3 scratch/egrad_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b11/egrad_check.txt
cat gpurun_out/b11/egrad_check.txt | tail -50
