#!/bin/bash
# does the round-2 tree show the one-lane sampler's nondeterminism on the 33-knot model?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/scratch/variants/wt_b
cp $GRAFT_REPO_ROOT/scratch/diag33b.py /tmp/diag33b.py
python3 /tmp/diag33b.py 2>&1 | grep -v amdgpu.ids | head -10
