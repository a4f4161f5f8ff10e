"""H psi of 2^20 walkers: shipped He model (k_efused<1>) and the 33-knot variant (k_efused<2>)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waveflow_amd.utils import physics
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
xb = bench.walkers(1 << 20, 4321).cuda()
m23, _ = bench.he_model("auto")
m33 = bench.seeded_model(2, 33, "auto")
for name, m in (("23 knots", m23), ("33 knots", m33)):
    rounds = [bench.event_ms(lambda: m.hamiltonian(xb, protons), 10, 5) for _ in range(3)]
    h = m.hamiltonian(xb, protons)
    h = h[0] if isinstance(h, (tuple, list)) else h
    print(f"{name}: {np.median(rounds):.4f} ms per 2^20 (rounds {['%.4f' % r for r in rounds]})  checksum {float(torch.as_tensor(h).double().sum()):.6e}", flush=True)
