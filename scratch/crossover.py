import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
from waveflow_amd import checkpoint, model_factory, _lib
flat = np.load('tests/golden/he_checkpoint.npz')['flat']
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10)
params, psi, log_pdf, sample = init_fun(0, 2)
m = log_pdf.model; m.set_params(flat)
L = _lib.lib()
g = np.random.default_rng(0)
for B in (256, 1024, 2048, 4096, 8192, 16384, 32768):
    x = torch.as_tensor(np.sort(g.uniform(-10, 10, size=(B, 2)), -1).astype(np.float32)).cuda()
    out = torch.empty(B, device='cuda')
    res = []
    for k in ("wave", "mfma"):
        m.set_kernel(k)
        sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda: L.wf_logpdf_fwd(m._h, ctypes.c_void_p(x.data_ptr()), B, ctypes.c_void_p(out.data_ptr()), None, None, sp)
        for _ in range(20): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 200
        for _ in range(n): f()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t) / n * 1e6)
    print(f"B={B}: wave {res[0]:.1f} us  mfma {res[1]:.1f} us")
