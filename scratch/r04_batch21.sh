#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b21; mkdir -p $O; rm -f $O/time.txt
for v in default eb55 prev_acc default eb55 prev_acc; do
  for smp in "" 1; do
    echo "== $v sampled=$smp" >> $O/time.txt
    if [ "$v" = "default" ]; then SAMPLED=$smp timeout 300 python3 scratch/r04_grad33_time.py 2>/dev/null | grep "23 knots matrix" >> $O/time.txt
    else SAMPLED=$smp WF_LIB=$PWD/scratch/variants/libwf_$v.so WF_LIB_EXPERIMENT=1 timeout 300 python3 scratch/r04_grad33_time.py 2>/dev/null | grep "23 knots matrix" >> $O/time.txt; fi
  done
done
cat $O/time.txt
