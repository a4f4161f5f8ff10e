#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scratch/diag33b.py 2>&1 | grep -v amdgpu.ids
POISON=0 python3 scratch/diag33b.py 2>&1 | grep -v amdgpu.ids
KNOTS=23 python3 scratch/diag33b.py 2>&1 | grep -v amdgpu.ids
