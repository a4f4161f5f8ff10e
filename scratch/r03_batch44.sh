#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b44
python3 scratch/crossover.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b44/crossover.txt
