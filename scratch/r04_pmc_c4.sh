#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over scratch/c4_prof.py -> gpurun_out/pmc_c4/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_c4
rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 scratch/c4_prof.py > $out/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_CYCLES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
WF_PMC_KERNELS=k_mfma python3 scratch/pmc_summary.py $out > $out/summary.txt 2>&1
find $out -name "*.csv" -delete; find $out -type d -empty -delete
cat $out/summary.txt
