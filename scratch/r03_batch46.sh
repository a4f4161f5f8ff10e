#!/bin/bash
# the round-2 tree's k_mfma with the SLP vectorizer switched back on (the build that failed in rounds 1 / 2: DESIGN 9) -- with lazily allocated (A) and
# pre-allocated (B) SGPR-spill VGPRs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/scratch/variants/wt_r2
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/b46
out=$GRAFT_REPO_ROOT/gpurun_out/b46/hazard_r2tree.txt
: > $out
for v in slpA slpB slpA slpB; do
  export WF_LIB=$PWD/scratch/variants/libwf_$v.so
  echo "== $v" >> $out
  REPS=16 CONFIGS="8x1 12x1 16x1" timeout 400 python3 scratch/hazard_probe.py 2>&1 | grep -v "amdgpu.ids" >> $out
done
cat $out
