#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b5
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/b5/tests.txt 2>&1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b5/bench_line.txt 2>&1
ROUNDS=2 CONFIGS="16x1" timeout 600 python3 scratch/time_ab.py default > gpurun_out/b5/time_default.txt 2>&1
tail -4 gpurun_out/b5/tests.txt; tail -1 gpurun_out/b5/bench_line.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('kernel_ms', d['roofline']['kernel_ms'], 'sustained', d.get('kernel_ms_sustained'))
for k in ('variant_33knot','c2_batch256_us','c4_d8_2pow18','rqs_2pow21','hpsi_2pow20','loss_grad_2pow17','nsc_2pow20'):
    print(k, json.dumps(d.get(k))[:300])
"; cat gpurun_out/b5/time_default.txt
