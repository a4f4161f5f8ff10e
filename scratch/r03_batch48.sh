#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b48
: > gpurun_out/b48/train_converge_seeds.txt
for seed in 3 4 5; do
  echo "== trainer seed $seed" >> gpurun_out/b48/train_converge_seeds.txt
  SEED=$seed CONFIGS="all grad" EPOCHS=20000 timeout 1500 python3 scratch/train_converge.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/b48/train_converge_seeds.txt
done
cat gpurun_out/b48/train_converge_seeds.txt
