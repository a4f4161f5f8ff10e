#!/bin/bash
# the reverse of a net as two launches (WF_GRAD_SPLIT=1): check against the wave sweeps + kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b19
export WF_GRAD_SPLIT=1
B=131072 timeout 600 python3 scratch/egrad_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b19/egrad_check.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b19/tmp -- python3 scratch/egrad_prof.py > gpurun_out/b19/prof.log 2>&1
find gpurun_out/b19/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b19/egrad_kernel_stats.csv; rm -rf gpurun_out/b19/tmp
grep -E "finite|vqmc|loss-grad" gpurun_out/b19/egrad_check.txt
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/b19/egrad_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('ewgrad','ebwd','efused','egrad_reduce')): print(r['Name'][27:64], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
