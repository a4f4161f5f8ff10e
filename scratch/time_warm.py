"""Kernel time of consecutive launches (He, 2^20 walkers) from a cold process: how long the GPU takes to reach its steady state."""
import os, sys, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("mfma")
x = bench.walkers(1 << 20, 1234).cuda()
m.log_pdf(x); torch.cuda.synchronize()
N = 3000
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
t0 = time.perf_counter()
for a, b in ev:
    a.record(); m.log_pdf(x); b.record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ts = np.array([a.elapsed_time(b) for a, b in ev])
print("wall %.3f s for %d launches" % (wall, N))
for lo, hi in ((0, 10), (10, 60), (60, 200), (200, 500), (500, 1000), (1000, 2000), (2000, 3000)):
    print("  launches %4d..%4d: mean %.4f ms  min %.4f" % (lo, hi, ts[lo:hi].mean(), ts[lo:hi].min()))
