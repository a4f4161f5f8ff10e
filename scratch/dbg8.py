import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
np.set_printoptions(linewidth=220, precision=6, suppress=True)
B = 1 << 20
dbg = torch.zeros(B * 2 * 16, device="cuda")
os.environ["WF_DBG_PTR"] = str(dbg.data_ptr())
import bench
m, flat = bench.he_model("scalar")
x = bench.walkers(B, 1234).cuda()
ls = m.log_pdf(x)
m.set_kernel("mfma")
shown = 0
for run in range(10):
    dbg.zero_()
    lm = m.log_pdf(x)
    bad = ((lm - ls).abs() > 0.05).nonzero().flatten()
    tiles = sorted(set((bad // 32).tolist()))
    for t in tiles:
        if shown >= 3: break
        g = dbg.view(B, 2, 16)[t * 32:(t + 1) * 32].cpu().numpy()
        names = ["t", "ynum", "dnum", "rS", "rs", "v0", "v15", "x", "il", "ir", "rl0", "rr0", "v7", "v8"]
        print("tile", t, "bad walkers", ((lm[t*32:(t+1)*32] - ls[t*32:(t+1)*32]).abs() > 0.05).nonzero().flatten().tolist())
        for k, nm in enumerate(names):
            a = g[:, :, k]
            if nm in ("il", "ir"): a = a.view(np.int32)
            print("  %-5s h0" % nm, a[:, 0]); print("  %-5s h1" % nm, a[:, 1])
        shown += 1
