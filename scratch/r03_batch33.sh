#!/bin/bash
# whole training steps at BASELINE config 5's payload per GPU (2^20 walkers)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b33
B=1048576 EPOCHS=20 timeout 900 python3 scratch/step_prof.py 2>&1 | grep -v amdgpu.ids | grep TILE_MIN=16384 > gpurun_out/b33/step20.txt
B=1048576 EPOCHS=60 timeout 900 python3 scratch/step_prof.py 2>&1 | grep -v amdgpu.ids | grep TILE_MIN=16384 > gpurun_out/b33/step60.txt
cat gpurun_out/b33/step20.txt gpurun_out/b33/step60.txt
