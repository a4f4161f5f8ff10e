#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b14; mkdir -p $O
EVENTS=1 timeout 600 python3 scratch/r04_stall_diag.py 2>/dev/null | tee $O/stall_events.txt
EVENTS=0 timeout 600 python3 scratch/r04_stall_diag.py 2>/dev/null | tee $O/stall_noevents.txt
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
