#!/bin/bash
# sampler occupancy: WF_OCC_SAMPLE = 3 (default), 4, 5 waves per SIMD
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WF_LIB_EXPERIMENT=1
for v in base occ4 occ5; do
  if [ $v = base ]; then unset WF_LIB; else export WF_LIB=$GRAFT_REPO_ROOT/scratch/variants/libwf_$v.so; fi
  echo "== $v"; python3 scratch/sample_prof.py 2>&1 | grep -E "sample|inverse"
done
