#!/bin/bash
: > gpurun_out/time.txt
for lib in "$@"; do
  if [ "$lib" = default ]; then unset WF_LIB; else export WF_LIB=$PWD/scratch/variants/libwf_$lib.so; fi
  timeout 300 python3 scratch/time_variant.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/time.txt
done
cat gpurun_out/time.txt
