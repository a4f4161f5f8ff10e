#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sample; mkdir -p $out
python3 scratch/sample_prof.py 2>&1 | grep -v amdgpu.ids
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 scratch/sample_prof.py > $out/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
WF_PMC_KERNELS=k_wave_sample python3 scratch/pmc_summary.py $out > $out/summary.txt 2>&1
find $out -name "*.csv" -size +2M -delete
cat $out/summary.txt
