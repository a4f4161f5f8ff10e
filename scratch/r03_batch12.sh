#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b12
B=131072 timeout 600 python3 scratch/egrad_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b12/egrad_check.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b12/tmp -- python3 scratch/egrad_prof.py > gpurun_out/b12/prof.log 2>&1
find gpurun_out/b12/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b12/egrad_kernel_stats.csv; rm -rf gpurun_out/b12/tmp
grep -E "finite|vqmc|loss-grad" gpurun_out/b12/egrad_check.txt; head -12 gpurun_out/b12/egrad_kernel_stats.csv | cut -c1-70,150-260
