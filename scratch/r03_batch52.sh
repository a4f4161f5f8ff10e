#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b52
timeout 1500 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_energy.py -m gpu -x -q -k "nonzero_boundary or local_energy or hamiltonian" > gpurun_out/b52/tests.txt 2>&1
tail -12 gpurun_out/b52/tests.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd.utils import physics
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
m, flat = bench.he_model("auto")
xb = bench.walkers(1 << 20, 4321).cuda()
print("hpsi 2^20: %.4f ms" % bench.event_ms(lambda: m.hamiltonian(xb, protons), 20, 10))
PY
