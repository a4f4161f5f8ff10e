#!/bin/bash
# A/B of experiment builds of the gradient path: kernel times of 6 loss-gradient calls on 2^17 walkers per variant (VARIANTS="base nt ...")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b16
export WF_LIB_EXPERIMENT=1
for v in ${VARIANTS:-base nt}; do
  if [ $v = base ]; then unset WF_LIB; else export WF_LIB=$GRAFT_REPO_ROOT/scratch/variants/libwf_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b16/tmp -- python3 scratch/egrad_prof.py > gpurun_out/b16/prof_$v.log 2>&1
  find gpurun_out/b16/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b16/stats_$v.csv; rm -rf gpurun_out/b16/tmp
  echo "== $v"; python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/b16/stats_$v.csv')):
    if any(k in r['Name'] for k in ('ewgrad','ebwd','efused','egrad_reduce')): print(r['Name'][27:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
done
