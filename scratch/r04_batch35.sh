#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b35; mkdir -p $O
for v in default fakeband; do
  if [ "$v" != "default" ]; then export WF_LIB=$PWD/scratch/variants/libwf_$v.so WF_LIB_EXPERIMENT=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$v -- python3 scratch/r04_sample_prof.py > $O/$v.log 2>&1
  find $O/tmp_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${v}_kernel_stats.csv; rm -rf $O/tmp_$v
  echo "== $v"; grep "k_tsample" $O/${v}_kernel_stats.csv | awk -F'",' '{n=split($0,a,","); print substr($1,2,60), a[n-4]}'
done
