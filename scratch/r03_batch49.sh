#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b49
timeout 1500 python3 scratch/outlier_diag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b49/outlier.txt
