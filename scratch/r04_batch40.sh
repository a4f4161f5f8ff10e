#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b40; mkdir -p $O
for mode in grouped onelane; do
  if [ "$mode" = "onelane" ]; then export WF_SAMPLE_ONE_LANE=1; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$mode -- python3 scratch/r04_train_trace.py > $O/trace_$mode.log 2>&1
  echo "== $mode"; python3 scratch/r04_trace_gaps.py $O/trace_$mode | head -12 | cut -c1-100
  rm -rf $O/trace_$mode
done | tee $O/timeline.txt
