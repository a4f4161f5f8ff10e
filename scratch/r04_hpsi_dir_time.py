"""H psi of 2^18 walkers through the directional matrix-core path, D = 8 and D = 4"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for D in (8, 4):
    m = bench.seeded_model(D, 23, "auto")
    x = bench.sorted_uniform(1 << 18, D, 1234).cuda()
    pr = np.linspace(-7.0, 7.0, D).astype(np.float32)
    rounds = [bench.event_ms(lambda: m.hamiltonian(x, pr), 5, 2) for _ in range(3)]
    h = m.hamiltonian(x, pr); h = h[0] if isinstance(h, (tuple, list)) else h
    print(f"D = {D}: {np.median(rounds):.3f} ms per 2^18 (rounds {['%.3f' % r for r in rounds]}) checksum {float(torch.as_tensor(h).double().abs().sum()):.6e}", flush=True)
