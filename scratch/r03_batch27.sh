#!/bin/bash
# kernel times of the staged sampler (sample_prof.py: sample, inverse, log_pdf of 2^17 walkers) and the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b27
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b27/tmp -- python3 scratch/sample_prof.py > gpurun_out/b27/prof.log 2>&1
find gpurun_out/b27/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b27/sampler_kernel_stats.csv; rm -rf gpurun_out/b27/tmp
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/b27/sampler_kernel_stats.csv')):
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
" | head -12
timeout 900 python3 bench.py > gpurun_out/b27/bench.txt 2>&1; tail -1 gpurun_out/b27/bench.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('value','ms_per_step','kernel_ms_sustained','hpsi_2pow20','hpsi_33knot_2pow20','loss_grad_2pow17','sample_2pow17','train_step_2pow17','c4_d8_2pow18','rqs_2pow21','nsc_2pow20'):
    print(k, d.get(k))
print('roofline', d['roofline']['frac'], d['roofline']['kernel_ms'])
"
