#!/bin/bash
: > gpurun_out/hazard3.txt
for k in "$@"; do
  WF_LIB=$PWD/scratch/variants/libwf_dbg${k}_fence.so timeout 200 python3 scratch/hazard_probe3.py ref /tmp/ref$k.pt 2>&1 | grep -v amdgpu.ids >> gpurun_out/hazard3.txt
  WF_LIB=$PWD/scratch/variants/libwf_dbg${k}_valu.so timeout 300 python3 scratch/hazard_probe3.py test /tmp/ref$k.pt 2>&1 | grep -v amdgpu.ids >> gpurun_out/hazard3.txt
done
cat gpurun_out/hazard3.txt
