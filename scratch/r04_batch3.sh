#!/bin/bash
# round 4, GPU batch 3: whole suite (-s: the measured gradient errors and parity lines go to the log), smoke, gradient path timing / kernel stats / PMC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b3; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q -s > $O/tests.txt 2>&1; echo "tests exit $?" >> $O/tests.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt
B=131072 timeout 600 python3 scratch/egrad_check.py 2>&1 | grep -v amdgpu.ids > $O/egrad_check.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp -- python3 scratch/egrad_prof.py > $O/prof.log 2>&1
find $O/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/grad_tile_kernel_stats.csv; rm -rf $O/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 scratch/egrad_prof.py > $O/pmc_$c.log 2>&1
done
WF_PMC_KERNELS=k_ebwd,k_efused,k_egrad python3 scratch/pmc_summary.py $O > $O/grad_tile_pmc.txt 2>&1; rm -rf $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE
grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^FAILED" $O/tests.txt; tail -4 $O/smoke.txt | cut -c1-300; grep -E "finite|vqmc|loss-grad" $O/egrad_check.txt; head -8 $O/grad_tile_kernel_stats.csv | cut -c1-70,150-260; cat $O/grad_tile_pmc.txt; grep "rel_l2 L" $O/tests.txt
