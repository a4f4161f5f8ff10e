"""Where do the sporadic ~40 ms stalls of the bench's timed region come from?  Repeats the 20-step region many times, timing every host call."""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waveflow_amd import _lib
dev = torch.device("cuda:0")
model, flat = bench.he_model("auto")
B = 1 << 20
x = bench.walkers(B, 1234).to(dev)
lp = torch.empty(B, device=dev)
L = _lib.lib()
ws = torch.empty(int(L.wf_block_sums_workspace_bytes(B)), device=dev, dtype=torch.uint8)
sums = torch.zeros(3, device=dev, dtype=torch.float64)
stream = torch.cuda.current_stream(dev)
sp = ctypes.c_void_p(stream.cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
use_events = os.environ.get("EVENTS", "1") == "1"
ev = [torch.cuda.Event(enable_timing=True) for _ in range(40)]
reg = []
worst = []
for rep in range(int(os.environ.get("REPS", 300))):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    tl = []
    for i in range(20):
        a = time.perf_counter()
        if use_events: ev[2 * i].record(stream)
        b = time.perf_counter()
        L.wf_logpdf_fwd(model._h, P(x), B, P(lp), None, None, sp)
        c = time.perf_counter()
        if use_events: ev[2 * i + 1].record(stream)
        d = time.perf_counter()
        L.wf_block_sums(P(lp), B, P(sums), P(ws), ws.numel(), sp)
        e = time.perf_counter()
        tl.append((b - a, c - b, d - c, e - d))
    f = time.perf_counter()
    torch.cuda.synchronize(dev)
    g = time.perf_counter()
    reg.append(g - t0)
    if g - t0 > 0.012:
        tl = np.array(tl) * 1e3
        worst.append((rep, (g - t0) * 1e3, (g - f) * 1e3, tl.max(0), tl.argmax(0)))
reg = np.array(reg) * 1e3
print(f"events={use_events}: regions {len(reg)}: median {np.median(reg):.3f} ms, p99 {np.percentile(reg, 99):.3f}, max {reg.max():.3f}; > 12 ms: {(reg > 12).sum()}")
for w in worst[:12]:
    print("  rep %d: region %.2f ms, final sync %.2f ms, worst host call (ev0, fwd, ev1, sums) ms %s at step %s" % (w[0], w[1], w[2], np.round(w[3], 2), w[4]))
