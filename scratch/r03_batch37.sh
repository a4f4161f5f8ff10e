#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scratch/diag33.py 2>&1 | grep -v amdgpu.ids
