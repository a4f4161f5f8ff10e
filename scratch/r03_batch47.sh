#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd.utils import physics
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
m33 = bench.seeded_model(2, 33, "auto")
x = m33.sample(3, 1 << 17, exact=True)
print("33 knots: loss+grad 2^17 %.3f ms (wave sweeps)" % bench.event_ms(lambda: m33.vqmc_loss_grad(x, protons, -1.8), 5, 2))
print("33 knots: H psi 2^17 %.3f ms" % bench.event_ms(lambda: m33.hamiltonian(x, protons), 10, 3))
print("33 knots: sample 2^17 %.3f ms" % bench.event_ms(lambda: m33.sample(5, 1 << 17, exact=True), 10, 3))
PY
