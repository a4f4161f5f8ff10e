#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 1 2 8; do WF_LIB=$PWD/scratch/variants/libwf_tsdbg$d.so WF_LIB_EXPERIMENT=1 timeout 300 python3 scratch/r04_p1g_diag2.py 2>/dev/null; done
