#!/bin/bash
# round 4, GPU batch 5: H psi beyond two particles (wf_kernels_etile_dir.hip): oracle / wave-kernel test at D = 3, 4, 8, timings, the bench line with the new legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b5; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_energy.py -q -s -k "beyond_two_particles or forward_laplacian" > $O/tests_dir.txt 2>&1; echo "exit $?" >> $O/tests_dir.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp -- python3 scratch/r04_hpsi_dir_prof.py > $O/prof.log 2>&1
find $O/tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/hpsi_dir_kernel_stats.csv; rm -rf $O/tmp
grep -E "hpsi D=|passed|failed|Error" $O/tests_dir.txt | tail -12; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04b5/bench_line.json') if l.startswith('{')][0])
for k in ("value","ms_per_step","c4_d8_2pow18","hpsi_c4_2pow18","hpsi_d4_2pow18","loss_grad_2pow17","train_step_2pow17","variant_33knot"):
    v=d.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ("kernels","roofline","workload","what")}
    print(k,v)
PY
head -8 $O/hpsi_dir_kernel_stats.csv | cut -c1-90,150-250
