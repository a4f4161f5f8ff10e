#!/bin/bash
# GPU-side: co-issue microbenchmark + hazard probe of every experiment build
mkdir -p gpurun_out
./scratch/ubench2/coissue.bin > gpurun_out/coissue.txt 2>&1
echo "coissue exit $?"
: > gpurun_out/hazard.txt
for lib in "" scratch/variants/libwf_fence_builtin.so scratch/variants/libwf_nofence_asm.so scratch/variants/libwf_nofence_builtin.so scratch/variants/libwf_nofence_bperm.so scratch/variants/libwf_nofence_asm_front.so; do
  if [ -n "$lib" ]; then export WF_LIB=$PWD/$lib; else unset WF_LIB; fi
  timeout 300 python3 scratch/hazard_probe.py >> gpurun_out/hazard.txt 2>&1
  echo "probe $lib exit $?" >> gpurun_out/hazard.txt
done
cat gpurun_out/coissue.txt
cat gpurun_out/hazard.txt
