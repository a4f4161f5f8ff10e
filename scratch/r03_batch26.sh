#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b26
timeout 1500 python3 -m pytest tests/test_gpu_inverse.py tests/test_gpu_grad.py -m gpu -x -q -k "staged or large_batch or samplers_draw or sampler_draws" > gpurun_out/b26/tests.txt 2>&1
tail -12 gpurun_out/b26/tests.txt
EPOCHS=60 timeout 900 python3 scratch/step_prof.py 2>&1 | grep -v amdgpu.ids | grep -E "TILE_MIN|sample" > gpurun_out/b26/step60.txt
EPOCHS=260 timeout 900 python3 scratch/step_prof.py 2>&1 | grep -v amdgpu.ids | grep TILE_MIN > gpurun_out/b26/step260.txt
cat gpurun_out/b26/step60.txt gpurun_out/b26/step260.txt
