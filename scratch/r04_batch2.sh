#!/bin/bash
# round 4, GPU batch 2: the dump-free gradient path (weight-gradient products inside k_ebwd): leaf-by-leaf check against the wave sweeps, the gradient
# tests of the suite, kernel stats and PMC (WRITE_SIZE / FETCH_SIZE) of a 2^17-walker call; A/B against the round-3 path (libwf_r04base.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b2; mkdir -p $O
timeout 900 python3 scratch/r04_parity_variants.py default nocenter xact xact_nocenter xsig xlog xrcp xall xall4 scalar > $O/parity_variants.txt 2>&1
timeout 900 python3 -m pytest tests -m gpu -q -k "antisymmetrised or checkpoint_artefacts or two_rank or fp16_range or staged_sampler_leaves or captured_large" > $O/tests_fixed.txt 2>&1; echo "exit $?" >> $O/tests_fixed.txt
B=131072 timeout 600 python3 scratch/egrad_check.py 2>&1 | grep -v amdgpu.ids > $O/egrad_check.txt
WF_LIB=$PWD/scratch/variants/libwf_r04base.so WF_LIB_EXPERIMENT=1 B=131072 timeout 600 python3 scratch/egrad_check.py 2>&1 | grep -v amdgpu.ids > $O/egrad_check_r03path.txt
timeout 900 python3 -m pytest tests/test_gpu_grad.py -q -x -k "matrix_cores or tile or large_batch or captured or deferred" > $O/tests_grad.txt 2>&1; echo "exit $?" >> $O/tests_grad.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp -- python3 scratch/egrad_prof.py > $O/prof.log 2>&1
find $O/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/grad_tile_kernel_stats.csv; rm -rf $O/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 scratch/egrad_prof.py > $O/pmc_$c.log 2>&1
done
WF_PMC_KERNELS=k_ebwd,k_efused,k_egrad python3 scratch/pmc_summary.py $O > $O/grad_tile_pmc.txt 2>&1; rm -rf $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE
cat $O/parity_variants.txt; tail -4 $O/tests_fixed.txt; grep -E "finite|vqmc|loss-grad" $O/egrad_check.txt; echo ---r03; grep -E "vqmc|loss-grad" $O/egrad_check_r03path.txt; tail -3 $O/tests_grad.txt; head -12 $O/grad_tile_kernel_stats.csv | cut -c1-70,150-260; cat $O/grad_tile_pmc.txt
