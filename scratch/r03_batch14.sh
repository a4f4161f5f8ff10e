#!/bin/bash
# k_ewgrad split count and the reverse kernel without its dump stores: kernel times of 6 loss-gradient calls on 2^17 walkers per variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b14
export WF_LIB_EXPERIMENT=1
for v in base s256 s512 nodump; do
  if [ $v = base ]; then unset WF_LIB; else export WF_LIB=$GRAFT_REPO_ROOT/scratch/variants/libwf_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b14/tmp -- python3 scratch/egrad_prof.py > gpurun_out/b14/prof_$v.log 2>&1
  find gpurun_out/b14/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b14/stats_$v.csv; rm -rf gpurun_out/b14/tmp
  echo "== $v"; head -8 gpurun_out/b14/stats_$v.csv | cut -c1-60,140-230
done
