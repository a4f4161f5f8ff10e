#!/bin/bash
# round 4, GPU batch 1: whole GPU suite (no -x: every failure in one pass), smoke tail (parity lines), A/B of the centred first hidden layer,
# driver-regime bench line + rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b1; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q -s > $O/tests.txt 2>&1; echo "tests exit $?" >> $O/tests.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt
ROUNDS=3 CONFIGS="16x1" timeout 600 python3 scratch/time_ab.py default nocenter > $O/time_ab.txt 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_driver_regime.log 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/driver_regime_kernel_stats.csv
rm -rf $O/stats
grep -E "passed|failed|error" $O/tests.txt | tail -5; tail -4 $O/smoke.txt; cat $O/time_ab.txt; head -3 $O/driver_regime_kernel_stats.csv
