#!/bin/bash
# does a smaller batch keep the tile dumps in the memory-side cache?  kernel times of the gradient path at 2^15, 2^16, 2^17 walkers (256 splits)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b15
export WF_LIB_EXPERIMENT=1
export WF_LIB=$GRAFT_REPO_ROOT/scratch/variants/libwf_s256.so
for b in 32768 65536 131072; do
  export B=$b
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b15/tmp -- python3 scratch/egrad_prof.py > gpurun_out/b15/prof_$b.log 2>&1
  find gpurun_out/b15/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b15/stats_$b.csv; rm -rf gpurun_out/b15/tmp
  echo "== $b"; python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/b15/stats_$b.csv')):
    if any(k in r['Name'] for k in ('ewgrad','ebwd','efused','egrad_reduce')): print(r['Name'][27:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
done
