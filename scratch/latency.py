import torch, time, sys, os, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd import _lib
L = _lib.lib()
m, flat = bench.he_model("auto")
for B in (256, 4096, 65536):
    x = bench.walkers(B, 1).cuda(); lp = torch.empty(B, device="cuda")
    sums = torch.zeros(3, device="cuda", dtype=torch.float64); ws = torch.empty(int(L.wf_block_sums_workspace_bytes(B)), device="cuda", dtype=torch.uint8)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    def step():
        sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.wf_logpdf_fwd(m._h, P(x), B, P(lp), None, None, sp); L.wf_block_sums(P(lp), B, P(sums), P(ws), ws.numel(), sp)
    for _ in range(20): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 200
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s): step()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t) / 200
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); L.wf_logpdf_fwd(m._h, P(x), B, P(lp), None, None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); ev1.record(); torch.cuda.synchronize()
    print(f"B={B}: step eager {eager*1e6:.1f} us, hipGraph replay {graph*1e6:.1f} us, log_pdf kernel alone {ev0.elapsed_time(ev1)*1e3:.1f} us")
