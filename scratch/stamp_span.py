import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
B = 1 << 20
W = int(os.environ.get("WF_MFMA_WAVES", "16"))
dbg = torch.zeros(256 * 16 * 8, device="cuda", dtype=torch.int64)
os.environ["WF_DBG_PTR"] = str(dbg.data_ptr())
import bench
m, flat = bench.he_model("mfma")
x = bench.walkers(B, 1234).cuda()
for _ in range(3): m.log_pdf(x)
dbg.zero_(); m.log_pdf(x); torch.cuda.synchronize()
g = dbg[:256 * W * 8].view(256, W, 8).cpu().numpy().astype(np.float64)
t0, t1, n = g[..., 0], g[..., 1], g[..., 2]
start = t0.min()
print("waves/WG %d: kernel span %.0f cycles (first start to last finish)" % (W, t1.max() - start))
for slot in range(W // 4):
    sel = slice(slot * 4, slot * 4 + 4)
    print("  SIMD slot %d: tiles per wave %.1f (min %d max %d)   finish at %.0f +- %.0f   start at %.0f" % (slot, n[:, sel].mean(), n[:, sel].min(), n[:, sel].max(), (t1[:, sel] - start).mean(), (t1[:, sel] - start).std(), (t0[:, sel] - start).mean()))
fin = (t1 - start).reshape(-1)
print("  finish time percentiles 1/25/50/75/99/100: " + " ".join("%.0f" % v for v in np.percentile(fin, [1, 25, 50, 75, 99, 100])))
print("  per workgroup: last finish - first finish: mean %.0f cycles; workgroup finish (max) mean %.0f, max over workgroups %.0f" % ((t1.max(1) - t1.min(1)).mean(), (t1.max(1) - start).mean(), (t1.max(1) - start).max()))
rt = g[..., 3]
wg_fin = rt.max(1)            # finish of each workgroup, 10 ns ticks
print("  workgroup finish on the constant clock: spread (max - min) %.1f us, std %.1f us; quartiles rel. to the first: %s us" % ((wg_fin.max() - wg_fin.min()) / 100.0, wg_fin.std() / 100.0, " ".join("%.1f" % ((v - wg_fin.min()) / 100.0) for v in np.percentile(wg_fin, [25, 50, 75, 90, 99]))))
