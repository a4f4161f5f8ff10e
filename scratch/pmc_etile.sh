#!/bin/bash
# PMC passes (separate runs, kernel-trace only, each under its own timeout) over scratch/etile_prof.py -> gpurun_out/pmc_etile/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_etile
rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 scratch/etile_prof.py > $out/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 scratch/pmc_summary_all.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
