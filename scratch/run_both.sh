#!/bin/bash
# hazard probe + timing of the default library (or of scratch/variants/libwf_$1.so)
if [ -n "$1" ] && [ "$1" != default ]; then export WF_LIB=$PWD/scratch/variants/libwf_$1.so; fi
REPS=${REPS:-20} timeout 400 python3 scratch/hazard_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/both.txt
timeout 300 python3 scratch/time_variant.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/both.txt
cat gpurun_out/both.txt
