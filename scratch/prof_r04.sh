#!/bin/bash
# round 4 profiles: the driver-regime bench line, kernel stats of the same command, PMC passes of the headline kernel (separate runs) -> gpurun_out/prof_r04/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r04
mkdir -p $out
stats() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tmp_$name -- "$@" > $out/$name.log 2>&1; find $out/tmp_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${name}_kernel_stats.csv; rm -rf $out/tmp_$name; }
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line_driver_regime.json 2> $out/bench_line_driver_regime.err
stats driver_regime python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras
bash scratch/pmc.sh r04 --no-extras > /dev/null 2>&1; cp gpurun_out/pmc_r04/summary.txt $out/pmc_summary.txt; rm -rf gpurun_out/pmc_r04
timeout 600 python3 bench.py --workload vqmc --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_line_vqmc.json 2>&1
timeout 900 python3 -m pytest tests -m gpu -q -x > $out/tests.txt 2>&1; echo "tests exit $?" >> $out/tests.txt
head -c 600 $out/bench_line_driver_regime.json; echo; head -4 $out/driver_regime_kernel_stats.csv | cut -c1-200; grep -E "FETCH|WRITE|INSTS_VALU |INSTS_MFMA|ACTIVE_INST_VALU|GRBM_GUI|TCC_HIT" $out/pmc_summary.txt; tail -3 $out/tests.txt
