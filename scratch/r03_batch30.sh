#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WF_LIB_EXPERIMENT=1 WF_LIB=$GRAFT_REPO_ROOT/scratch/variants/libwf_tscount.so
python3 scratch/ts_count.py 2>&1 | grep -v amdgpu.ids
