#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b20
EPOCHS=60 timeout 900 python3 scratch/step_prof.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b20/step60.txt
EPOCHS=260 timeout 900 python3 scratch/step_prof.py 2>&1 | grep -v amdgpu.ids | grep TILE_MIN > gpurun_out/b20/step260.txt
cat gpurun_out/b20/step60.txt gpurun_out/b20/step260.txt
