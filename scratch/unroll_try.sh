#!/bin/bash
for cfg in "-DWF_GEMV_UNROLL_BWD=8" "-DWF_GEMV_UNROLL_BWD=16" "-DWF_GEMV_UNROLL_BWD=2"; do
  touch waveflow_amd/csrc/wf_kernels_wave.hip
  WF_CXXFLAGS="$cfg" python -m waveflow_amd.build > /dev/null 2>&1
  echo "=== flags: $cfg"
  python scratch/grad_ab.py 2 4 2>&1 | grep "^D=" | cut -c1-75
  python scratch/bench_grad.py 2>&1 | grep "B=131072\|B=256"
  python scratch/bench_mle.py 2>&1 | grep -E "MFlow" | cut -c1-80
done
