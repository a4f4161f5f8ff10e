#!/bin/bash
# PMC passes (separate runs, kernel-trace only) of the gradient path's kernels: 6 loss-gradient calls on 2^17 walkers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_egrad
mkdir -p $out
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 scratch/egrad_prof.py > $out/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT
run sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES
run sq4 SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_ITEMS
run fetch FETCH_SIZE
run write WRITE_SIZE
WF_PMC_KERNELS=k_ebwd,k_ewgrad,k_efused python3 scratch/pmc_summary.py $out > $out/summary.txt 2>&1
find $out -name "*.csv" -size +2M -delete
cat $out/summary.txt
