import numpy as np, torch, sys, os, time, ctypes
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd import _lib
L = _lib.lib()
model, flat = bench.he_model("mfma")
B = 1 << 20
x = bench.walkers(B, 1234).cuda()
lp = torch.empty(B, device="cuda")
ws = torch.empty(int(L.wf_block_sums_workspace_bytes(B)), device="cuda", dtype=torch.uint8)
sums = torch.zeros(3, device="cuda", dtype=torch.float64)
P = lambda t: ctypes.c_void_p(t.data_ptr())
stream = torch.cuda.current_stream()
sp = ctypes.c_void_p(stream.cuda_stream)
def run(n, with_sums, with_events):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stamps = []
    for i in range(n):
        if with_events: evs[i][0].record(stream)
        L.wf_logpdf_fwd(model._h, P(x), B, P(lp), None, None, sp)
        if with_events: evs[i][1].record(stream)
        if with_sums: L.wf_block_sums(P(lp), B, P(sums), P(ws), ws.numel(), sp)
        stamps.append(time.perf_counter() - t0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    k = np.mean([a.elapsed_time(b) for a, b in evs]) if with_events else float('nan')
    print(f"n={n} sums={with_sums} events={with_events}: host issue {(t1-t0)*1e3:.2f} ms total {(t2-t0)*1e3:.2f} ms => {(t2-t0)*1e3/n:.3f} ms/step; kernel_ms(ev) {k:.3f}; max host step {np.max(np.diff(stamps))*1e3:.3f} ms")
for rep in range(2):
    run(20, False, False); run(20, True, False); run(20, False, True); run(20, True, True); run(200, True, True)
