"""k_mfma launch time against the batch size: slope = per-tile cost, intercept = launch + LDS staging of the operand images."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
m, flat = bench.he_model("mfma")
res = []
for logb in (15, 16, 17, 18, 19, 20, 21):
    B = 1 << logb
    x = bench.walkers(B, 1234).cuda()
    for _ in range(10): m.log_pdf(x)
    ts = []
    for _ in range(60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.log_pdf(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    res.append((B, np.median(ts), np.min(ts)))
    print(f"B = 2^{logb}: median {np.median(ts)*1e3:8.1f} us   min {np.min(ts)*1e3:8.1f} us   per 2^20: {np.median(ts) * (1 << 20) / B:.4f} ms")
B = np.array([r[0] for r in res[2:]], float); t = np.array([r[2] for r in res[2:]])
a, b = np.polyfit(B, t, 1)
print(f"fit (min times, B >= 2^17): {a * (1 << 20) * 1e3:.1f} us per 2^20 walkers + {b * 1e3:.1f} us")
