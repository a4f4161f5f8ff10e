#!/bin/bash
# round 3 profiles: kernel stats of the driver-regime bench command and of the secondary legs, PMC passes (separate runs) -> gpurun_out/prof_r03/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03
mkdir -p $out
stats() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tmp_$name -- "$@" > $out/$name.log 2>&1; find $out/tmp_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${name}_kernel_stats.csv; rm -rf $out/tmp_$name; }
pmc() { name=$1; ctr=$2; shift 2; timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$name/$(echo $ctr | tr ' ' '_') -- "$@" > /dev/null 2>&1; }
stats driver_regime python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras
stats rqs python3 bench.py --workload rqs --steps 50 --warmup 20
stats energy python3 scratch/etile_prof.py
stats loss_grad python3 scratch/grad_prof.py
stats nsc python3 bench.py --workload nsc --steps 30 --warmup 10
for c in FETCH_SIZE WRITE_SIZE; do pmc rqs $c python3 bench.py --workload rqs --steps 5 --warmup 2; pmc energy $c python3 scratch/etile_prof.py; done
WF_PMC_KERNELS=k_rqs python3 scratch/pmc_summary.py $out/pmc_rqs > $out/rqs_pmc.txt 2>&1
WF_PMC_KERNELS=k_efused,k_etile python3 scratch/pmc_summary.py $out/pmc_energy > $out/energy_pmc.txt 2>&1
rm -rf $out/pmc_rqs $out/pmc_energy
bash scratch/pmc.sh r03 --no-extras > /dev/null 2>&1; cp gpurun_out/pmc_r03/summary.txt $out/pmc_summary.txt; rm -rf gpurun_out/pmc_r03
for f in $out/*_kernel_stats.csv; do echo == $f; head -4 $f | cut -c1-200; done; cat $out/rqs_pmc.txt $out/energy_pmc.txt
