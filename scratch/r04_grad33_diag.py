"""leaf-by-leaf comparison of the matrix-core gradient path (two row blocks) with the wave sweeps"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from waveflow_amd import checkpoint, model_factory
from test_gpu_grad import sorted_walkers, _leaves
kn = int(os.environ.get("KN", 33))
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=kn, n_i_internal_knots=kn,
                                            i_spline_reg=0.05, n_flow_layers=int(os.environ.get("NL", 3)), box_size=10.0)
params, psi, log_pdf, sample = init_fun(3, 2)
m = psi.model
m.ensure_params(params)
B = int(os.environ.get("B", 171))
x = sorted_walkers(B, 2, 9.0, 21)
g = np.random.default_rng(8)
wp, wl = g.normal(size=B).astype(np.float32), (float(os.environ.get("WL", 0.1)) * g.normal(size=B)).astype(np.float32)
os.environ["WF_GRAD_TILE_MIN"] = "1"
t = m.psi_vjp(x, wp, wl).cpu().numpy().astype(np.float64)
os.environ["WF_GRAD_TILE_MIN"] = "0"
w = m.psi_vjp(x, wp, wl).cpu().numpy().astype(np.float64)
print("overall rel", np.linalg.norm(t - w) / np.linalg.norm(w))
for i, (lt, lw) in enumerate(zip(_leaves(checkpoint.unflatten_like(params, t)), _leaves(checkpoint.unflatten_like(params, w)))):
    if lw.size == 0: continue
    err = np.linalg.norm(lt - lw) / max(np.linalg.norm(lw), 1e-30)
    print(i, lw.shape, f"rel {err:.2e}  |w| {np.linalg.norm(lw):.3e}")
    if err > 1e-3 and lw.ndim == 2 and lw.shape[1] > 64:
        d = np.abs(lt - lw)
        cols = d.max(0)
        print("   worst columns", np.argsort(cols)[-8:], cols[np.argsort(cols)[-8:]], "col norms of w", np.abs(lw).max(0)[np.argsort(cols)[-8:]])
    if err > 1e-3 and lw.ndim == 1:
        d = np.abs(lt - lw); idx = np.argsort(d)[-8:]
        print("   worst", idx, lt[idx], lw[idx])
