#!/bin/bash
# PMC passes of the 8-electron chain's log_pdf kernel (k_mfma<8,1,8,1>)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_c4; mkdir -p $out
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 scratch/c4_prof.py > $out/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
WF_PMC_KERNELS=k_mfma python3 scratch/pmc_summary.py $out > $out/summary.txt 2>&1
find $out -name "*.csv" -size +2M -delete
cat $out/summary.txt
