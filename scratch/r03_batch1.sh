#!/bin/bash
# round 3, GPU batch 1: test suite, VALU issue microbenchmark, A/B timing of k_mfma experiment builds, driver-regime kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b1
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/b1/tests.txt 2>&1; echo "tests exit $?" >> gpurun_out/b1/tests.txt
timeout 200 scratch/ubench2/valu_mix.bin > gpurun_out/b1/valu_mix.txt 2>&1
ROUNDS=3 CONFIGS="16x1 12x1 8x1" timeout 1200 python3 scratch/time_ab.py default pf trunc pftrunc p4 > gpurun_out/b1/time_ab.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/b1/bench_driver_regime.log 2>&1
find gpurun_out/b1/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/b1/driver_regime_kernel_stats.csv
rm -rf gpurun_out/b1/stats
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b1/bench_driver_line.txt 2>&1
tail -3 gpurun_out/b1/tests.txt; cat gpurun_out/b1/valu_mix.txt; cat gpurun_out/b1/time_ab.txt; head -4 gpurun_out/b1/driver_regime_kernel_stats.csv
