"""soak: the ticketed two-row-block gradient path and the eight-lane sampler kernels, thousands of repeats: every result equal to the first one"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from test_gpu_grad import sorted_walkers
from waveflow_amd.utils import physics
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
n_rep = int(os.environ.get("REPS", 2000))
for kn in (33, 23):
    m = bench.seeded_model(2, kn, "auto") if kn == 33 else bench.he_model("auto")[0]
    g = np.random.default_rng(8)
    for B in (50001, 1 << 17):
        xb = torch.as_tensor(sorted_walkers(B, 2, 9.5, 7)).cuda()
        w1 = torch.as_tensor(g.normal(size=B).astype(np.float32)).cuda()
        w2 = torch.as_tensor((0.1 * g.normal(size=B)).astype(np.float32)).cuda()
        first = m.psi_vjp(xb, w1, w2)
        t0 = time.time(); bad = 0
        for i in range(n_rep):
            if not torch.equal(m.psi_vjp(xb, w1, w2), first): bad += 1
        print(f"knots {kn} B {B}: psi_vjp x {n_rep}: {bad} differ from the first ({time.time() - t0:.1f} s)", flush=True)
    first = m.sample(5, 1 << 17, exact=True)
    bad = sum(0 if torch.equal(m.sample(5, 1 << 17, exact=True), first) else 1 for _ in range(n_rep // 4))
    print(f"knots {kn}: sample x {n_rep // 4}: {bad} differ from the first", flush=True)
