#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b8; mkdir -p $O
timeout 300 python3 scratch/r04_grad33_diag.py > $O/diag33.txt 2>&1
KN=23 timeout 300 python3 scratch/r04_grad33_diag.py > $O/diag23.txt 2>&1
WL=0 timeout 300 python3 scratch/r04_grad33_diag.py > $O/diag33_wl0.txt 2>&1
tail -40 $O/diag33.txt; tail -30 $O/diag23.txt; tail -30 $O/diag33_wl0.txt
