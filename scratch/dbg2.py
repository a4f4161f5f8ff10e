import numpy as np, torch, sys, os, time, ctypes
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd import _lib
L = _lib.lib()
for kern in ("mfma", "scalar"):
    model, flat = bench.he_model(kern)
    B = 1 << 20
    x = bench.walkers(B, 1234).cuda()
    lp = torch.empty(B, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(3):
        L.wf_logpdf_fwd(model._h, P(x), B, P(lp), None, None, sp)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(20):
            L.wf_logpdf_fwd(model._h, P(x), B, P(lp), None, None, sp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(kern, "20 launches: host issue %.3f ms, total %.3f ms => %.3f ms/launch" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / 20))
