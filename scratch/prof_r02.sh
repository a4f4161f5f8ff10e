#!/bin/bash
# rocprofv3 kernel stats of the bench command + PMC passes (separate runs), summaries under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r02
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02/stats -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_r02/bench_stats.log 2>&1
find gpurun_out/prof_r02/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_r02/kernel_stats.csv
bash scratch/pmc.sh r02c --no-extras > /dev/null 2>&1
cp gpurun_out/pmc_r02c/summary.txt gpurun_out/prof_r02/pmc_summary.txt
head -5 gpurun_out/prof_r02/kernel_stats.csv
cat gpurun_out/prof_r02/pmc_summary.txt
tail -2 gpurun_out/prof_r02/bench_stats.log | cut -c1-400
