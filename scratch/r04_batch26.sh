#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b26; mkdir -p $O
timeout 600 python3 scratch/r04_repro_diag.py 2>/dev/null | tee $O/repro.txt
