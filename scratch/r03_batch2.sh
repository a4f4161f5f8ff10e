#!/bin/bash
# round 3, GPU batch 2: why is k_mfma as fast at 8 waves per workgroup as at 16?  PMC passes and per-phase stamps at both shapes; parity log lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b2
rocprofv3 -L > gpurun_out/b2/counters.txt 2>&1
for w in 8 16; do
  WF_MFMA_WAVES=$w bash scratch/pmc.sh b2w$w --no-extras > /dev/null 2>&1
  cp gpurun_out/pmc_b2w$w/summary.txt gpurun_out/b2/pmc_waves$w.txt
  rm -rf gpurun_out/pmc_b2w$w
  WF_MFMA_WAVES=$w WF_LIB=$PWD/scratch/variants/libwf_stamp.so WF_LIB_EXPERIMENT=1 timeout 300 python3 scratch/stamp.py > gpurun_out/b2/stamp_waves$w.txt 2>&1
done
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "strict or full_size or general_boundary or gated_conditioner" > gpurun_out/b2/parity_lines.txt 2>&1
grep -E "^\[(strict|parity)" gpurun_out/b2/parity_lines.txt > gpurun_out/b2/parity_lines_short.txt
cat gpurun_out/b2/pmc_waves8.txt gpurun_out/b2/pmc_waves16.txt gpurun_out/b2/stamp_waves*.txt; tail -3 gpurun_out/b2/parity_lines.txt
