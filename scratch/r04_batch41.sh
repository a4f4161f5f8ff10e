#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b41; mkdir -p $O
timeout 900 python3 scratch/r04_soak.py 2>/dev/null | tee $O/soak.txt
