import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("mfma")
B = 1 << 20
x = bench.walkers(B, 1234).cuda()
res = {}
for k in ("scalar", "mfma"):
    m.set_kernel(k)
    lp, u = m.log_pdf(x, return_sample=True)
    ps = m.psi(x)
    res[k] = (lp.cpu().numpy().astype(np.float64), ps.cpu().numpy().astype(np.float64), u.cpu().numpy())
ls, pss, us = res["scalar"]; lm, pm, um = res["mfma"]
d = np.abs(lm - ls); o = np.argsort(-d)[:8]
print("max |lp mfma - scalar|", d.max(), "n>1e-2:", (d > 1e-2).sum(), "n>1e-3", (d > 1e-3).sum())
xs = x.cpu().numpy()
for i in o: print(i, xs[i], "lp", ls[i], lm[i], "psi", pss[i], pm[i], "u", us[i], um[i])
dp = np.abs(pm - pss); print("max |psi diff|", dp.max(), np.argmax(dp))
m.set_kernel("mfma")
lp2 = m.log_pdf(x).cpu().numpy(); print("mfma rerun identical:", np.array_equal(lp2, lm.astype(np.float32)))
