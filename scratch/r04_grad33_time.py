"""loss + gradient of 2^17 walkers: the shipped He model (23 knots, one row block) and the 33-knot variant (two row blocks) on the matrix-core
gradient path, the 33-knot variant also on the wave sweeps (WF_GRAD_TILE_MIN=0)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waveflow_amd.utils import physics
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
xg = bench.walkers(1 << 17, 1234).cuda()
m23, _ = bench.he_model("auto")
m33 = bench.seeded_model(2, 33, "auto")
if os.environ.get("SAMPLED"):   # walkers from the shipped model's own |psi|^2 (the training step's input) instead of uniform ones
    xg = m23.sample(11, 1 << 17, exact=True)
for name, m in (("23 knots", m23), ("33 knots", m33)):
    for tile_min in (None, "0"):
        if tile_min is None: os.environ.pop("WF_GRAD_TILE_MIN", None)
        else: os.environ["WF_GRAD_TILE_MIN"] = tile_min
        rounds = [bench.event_ms(lambda: m.vqmc_loss_grad(xg, protons, -1.8), 10, 3) for _ in range(3)]
        s, g = m.vqmc_loss_grad(xg, protons, -1.8)
        print(f"{name} {'matrix cores' if tile_min is None else 'wave sweeps '}: {np.median(rounds):.3f} ms per 2^17 (rounds {['%.3f' % r for r in rounds]})  |grad| {float(g.double().norm()):.6e}", flush=True)
os.environ.pop("WF_GRAD_TILE_MIN", None)
