#!/bin/bash
# end-of-round evidence for the gradient path as it now stands: check against the wave sweeps, kernel stats, HBM traffic (23 and 33 knots), training convergence, vqmc bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b20; mkdir -p $O
B=131072 timeout 600 python3 scratch/egrad_check.py 2>&1 | grep -v amdgpu.ids > $O/egrad_check.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp -- python3 scratch/egrad_prof.py > $O/prof.log 2>&1
find $O/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/grad_tile_kernel_stats.csv; rm -rf $O/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 scratch/egrad_prof.py > $O/pmc_$c.log 2>&1
done
WF_PMC_KERNELS=k_ebwd,k_efused,k_egrad python3 scratch/pmc_summary.py $O > $O/grad_tile_pmc.txt 2>&1; rm -rf $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE
timeout 600 python3 bench.py --workload vqmc --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_line_vqmc.json 2>$O/bench_vqmc.err

grep -E "finite|vqmc|loss-grad" $O/egrad_check.txt; head -7 $O/grad_tile_kernel_stats.csv | cut -c1-70,150-260; cat $O/grad_tile_pmc.txt; tail -c 600 $O/bench_line_vqmc.json
