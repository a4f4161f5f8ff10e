"""staged sampler, 2^17 walkers: shipped He model and the 33-knot variant"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
m23, _ = bench.he_model("auto")
m33 = bench.seeded_model(2, 33, "auto")
for name, m in (("23 knots", m23), ("33 knots", m33)):
    seeds = iter(range(100, 100000))
    rounds = [bench.event_ms(lambda: m.sample(next(seeds), 1 << 17, exact=True), 10, 3) for _ in range(3)]
    x = m.sample(7, 1 << 17, exact=True)
    print(f"{name}: {np.median(rounds):.4f} ms per 2^17 draws (rounds {['%.4f' % r for r in rounds]})  checksum {float(x.double().sum()):.6f} {float(x.double().abs().sum()):.6f}", flush=True)
