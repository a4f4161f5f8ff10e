#!/bin/bash
: > gpurun_out/hazard2.txt
for lib in "$@"; do
  export WF_LIB=$PWD/scratch/variants/libwf_$lib.so
  REPS=${REPS:-30} timeout 300 python3 scratch/hazard_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/hazard2.txt
done
grep "waves" gpurun_out/hazard2.txt
