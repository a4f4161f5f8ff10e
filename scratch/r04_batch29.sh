#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b29; mkdir -p $O; rm -f $O/time.txt
for v in default ilp3 default ilp3; do
  echo "== $v" >> $O/time.txt
  if [ "$v" = "default" ]; then timeout 300 python3 scratch/r04_hpsi_time.py 2>/dev/null >> $O/time.txt
  else WF_LIB=$PWD/scratch/variants/libwf_$v.so WF_LIB_EXPERIMENT=1 timeout 300 python3 scratch/r04_hpsi_time.py 2>/dev/null >> $O/time.txt; fi
done
cat $O/time.txt

