#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b10
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c4 or reproducible or strict or other_dim or shapes" > gpurun_out/b10/tests.txt 2>&1
timeout 300 python3 - > gpurun_out/b10/c4_time.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
m8 = bench.seeded_model(8, 23, "mfma")
x8 = bench.sorted_uniform(1 << 18, 8, 1234).cuda()
for rep in range(3):
    print("C4 2^18: %.4f ms" % bench.kernel_ms(m8, x8, n=20, warm=10))
x8b = bench.sorted_uniform(1 << 20, 8, 1234).cuda()
print("C4 2^20: %.4f ms" % bench.kernel_ms(m8, x8b, n=10, warm=5))
PY
tail -4 gpurun_out/b10/tests.txt; grep C4 gpurun_out/b10/c4_time.txt
