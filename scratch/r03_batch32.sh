#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b32
timeout 900 python3 scratch/tsample_ks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b32/ks.txt
