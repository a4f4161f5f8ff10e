#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b43
timeout 1500 python3 -m pytest tests/test_gpu_inverse.py -m gpu -x -q > gpurun_out/b43/tests.txt 2>&1
tail -15 gpurun_out/b43/tests.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
m33 = bench.seeded_model(2, 33, "auto")
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for tm in ("16384", "0"):
    os.environ["WF_SAMPLE_TILE_MIN"] = tm
    print("33 knots, WF_SAMPLE_TILE_MIN", tm, ": sample 2^17 %.3f ms" % (t(lambda: m33.sample(5, 1 << 17, exact=True)) * 1e3))
PY
