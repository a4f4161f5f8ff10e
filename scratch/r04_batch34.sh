#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b34; rm -rf $O; mkdir -p $O
pm() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/p/$name -- python3 scratch/r04_sample_prof.py > $O/$name.log 2>&1; }
pm sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES
pm sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_FLAT
pm tcc TCC_HIT_sum TCC_MISS_sum
WF_PMC_KERNELS=k_tsample python3 scratch/pmc_summary.py $O/p > $O/summary.txt 2>&1; rm -rf $O/p
cat $O/summary.txt
