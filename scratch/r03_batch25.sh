#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b25
timeout 900 python3 scratch/tsample_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b25/tsample_check.txt
