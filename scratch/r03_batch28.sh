#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b28
timeout 900 python3 scratch/tsample_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b28/tsample_check.txt
timeout 1500 python3 -m pytest tests/test_gpu_inverse.py tests/test_gpu_grad.py -m gpu -x -q -k "staged or large_batch or samplers_draw or sampler_draws" > gpurun_out/b28/tests.txt 2>&1
tail -5 gpurun_out/b28/tests.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b28/tmp -- python3 scratch/sample_prof.py > gpurun_out/b28/prof.log 2>&1
find gpurun_out/b28/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b28/sampler_kernel_stats.csv; rm -rf gpurun_out/b28/tmp
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/b28/sampler_kernel_stats.csv')):
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
" | head -5
