import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("scalar")
B = 1 << 20
x = bench.walkers(B, 1234).cuda()
ls = m.log_pdf(x)
m.set_kernel("mfma")
outs = [m.log_pdf(x) for _ in range(6)]
for i, o in enumerate(outs):
    d = (o - ls).abs()
    bad = (d > 0.05).nonzero().flatten()
    print("run", i, "n(|mfma-scalar|>0.05) =", bad.numel(), "tiles:", sorted(set((bad // 32).tolist()))[:10], "equal to run0:", torch.equal(o, outs[0]))
