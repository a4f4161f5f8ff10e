#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WF_LIB_EXPERIMENT=1 WF_LIB=$GRAFT_REPO_ROOT/scratch/variants/libwf_prealloc.so
for i in 1 2 3; do python3 scratch/diag33b.py 2>&1 | grep -v "amdgpu.ids\|experiment lib" | grep -E "call|nan count|finite in both"; done
