#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b24; rm -rf $O; mkdir -p $O
export KN=23
pm() { tag=$1; name=$2; shift 2; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag/$name -- python3 scratch/r04_grad33_prof.py > $O/$tag.$name.log 2>&1; }
both() { tag=$1
  pm $tag sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
  pm $tag sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pm $tag sq3 SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_IFETCH
  WF_PMC_KERNELS=k_ebwd python3 scratch/pmc_summary.py $O/$tag > $O/summary_$tag.txt 2>&1
}
both new
export WF_LIB=$PWD/scratch/variants/libwf_e1swap.so WF_LIB_EXPERIMENT=1
both old
rm -rf $O/new $O/old
paste <(grep -A40 "k_ebwd<false" $O/summary_new.txt | head -32) <(grep -A40 "k_ebwd<false" $O/summary_old.txt | head -32 | cut -c30-60)
