import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
B = 1 << 20
dbg = torch.zeros(256 * 16 * 8, device="cuda", dtype=torch.int64)
os.environ["WF_DBG_PTR"] = str(dbg.data_ptr())
import bench
m, flat = bench.he_model("mfma")
x = bench.walkers(B, 1234).cuda()
for _ in range(3): m.log_pdf(x)
dbg.zero_(); m.log_pdf(x); torch.cuda.synchronize()
W = int(os.environ.get("WF_MFMA_WAVES", "12"))
g = dbg[:256 * W * 8].view(256, W, 8).cpu().numpy().astype(np.float64)
tiles_per_wave = (B // 32) / (256 * W)
names = ["0 pre-layer (box/prev tail)", "1 hidden layers", "2 dim0 block", "3 out_block d=1 (MFMA)", "4 sigmoid block", "5 lerp d=1", "6 prior+store"]
tot = g.sum(-1).mean()
print("waves/WG", W, "tiles/wave %.2f" % tiles_per_wave, "total stamped cycles per wave %.0f => per tile %.0f" % (tot, tot / tiles_per_wave))
for k, nm in enumerate(names):
    print("  %-30s %9.0f cycles/tile  (%.1f%%)" % (nm, g[:, :, k].mean() / tiles_per_wave, 100 * g[:, :, k].mean() / tot))
