#!/bin/bash
# the reference's example script at a large batch (default reference-mode sampler and the exact one), checkpoints included
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b50
( time timeout 600 python3 examples/run_vqmc.py --epochs 300 --batch 65536 --lr 1e-3 --log-every 100 --save-dir /tmp/wf_ex_a ) 2>&1 | grep -v amdgpu.ids | tail -8
( time timeout 600 python3 examples/run_vqmc.py --epochs 300 --batch 65536 --lr 1e-3 --log-every 100 --exact-sampler --save-dir /tmp/wf_ex_b ) 2>&1 | grep -v amdgpu.ids | tail -8
ls /tmp/wf_ex_b | head
