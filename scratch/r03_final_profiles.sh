#!/bin/bash
# end-of-round records with the final build: the driver-regime bench line and its kernel stats, the default bench line, kernel stats of the gradient and sampler paths
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
stats() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tmp_$name -- "$@" > $out/$name.log 2>&1; find $out/tmp_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${name}_kernel_stats.csv; rm -rf $out/tmp_$name; }
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_regime.json 2> $out/bench_driver_regime.err
stats driver_regime python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras
stats grad_tile python3 scratch/egrad_prof.py
stats staged_sampler python3 scratch/sample_prof.py
timeout 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 600 python3 __graft_entry__.py --smoke > $out/smoke.txt 2>&1
for f in $out/*_kernel_stats.csv; do echo == $f; head -5 $f | cut -c1-160; done
tail -c 600 $out/bench_driver_regime.json; echo; tail -3 $out/smoke.txt
