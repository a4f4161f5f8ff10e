#!/bin/bash
# does the sporadic ~80 ms host-side stall in the loss+gradient loop come with kernels that use scratch (register spills)?
for cfg in "" "-DWF_OCC_FWD2=2"; do
  touch waveflow_amd/csrc/wf_kernels_wave.hip
  WF_CXXFLAGS="$cfg" python -m waveflow_amd.build > /dev/null 2>&1
  echo "=== flags: $cfg"
  for i in 1 2 3 4 5 6 7 8; do python scratch/bench_grad.py 2>&1 | grep "B=32768" | cut -c1-40; done
done
