#!/bin/bash
# the round-1 corruption of k_mfma (SLP-vectorised, packed-FP32 build) with lazily allocated (A) and pre-allocated (B) SGPR-spill VGPRs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b41
export WF_LIB_EXPERIMENT=1
for v in slpA slpB slpA slpB; do
  export WF_LIB=$GRAFT_REPO_ROOT/scratch/variants/libwf_$v.so
  echo "== $v" >> gpurun_out/b41/hazard.txt
  REPS=16 CONFIGS="8x1 12x1 16x1" timeout 400 python3 scratch/hazard_probe.py 2>&1 | grep -v "amdgpu.ids\|experiment lib" >> gpurun_out/b41/hazard.txt
done
cat gpurun_out/b41/hazard.txt
