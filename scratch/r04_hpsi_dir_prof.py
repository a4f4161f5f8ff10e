"""rocprofv3 target: H psi of 2^18 walkers of the 8-electron chain (C4's model) and of a 4-electron chain through the directional matrix-core path."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
for D in (8, 4):
    m = bench.seeded_model(D, 23, "auto")
    x = bench.sorted_uniform(1 << 18, D, 1234).cuda()
    pr = np.linspace(-7.0, 7.0, D).astype(np.float32)
    for _ in range(4):
        m.hamiltonian(x, pr)
    torch.cuda.synchronize()
