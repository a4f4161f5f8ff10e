#!/bin/bash
# gradient path: check + kernel times, then the gradient tests (incl. the other-models test)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scratch/r03_batch12.sh
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/b12/egrad_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('ewgrad','ebwd','efused','egrad_reduce')): print(r['Name'][27:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
mkdir -p gpurun_out/b18
timeout 1500 python3 -m pytest tests/test_gpu_grad.py -m gpu -x -q -k "matrix_cores or tile_path or psi_vjp or vqmc_loss" > gpurun_out/b18/tests.txt 2>&1
tail -15 gpurun_out/b18/tests.txt
