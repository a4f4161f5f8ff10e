import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory, flatten_params
sys.path.insert(0, 'tests')
from conftest import sorted_walkers
D=3
init_fun = model_factory.get_waveflow_model(D, base_spline_degree=4, i_spline_degree=4, n_prior_internal_knots=13, n_i_internal_knots=13, i_spline_reg=0.05, n_flow_layers=2, box_size=5.0, xu_coord_type="first")
params, psi, log_pdf, sample = init_fun(7, D)
m = psi.model
m.ensure_params(params)
x = sorted_walkers(37, D, 4.5, 21)
for k in ("scalar", "mfma"):
    m.set_kernel(k)
    print(k, "logpdf nan:", np.isnan(m.log_pdf(x)).sum(), "psi nan:", np.isnan(m.psi(x)).sum())
h, ps, lap = m.hamiltonian(x, [0.0], return_psi=True, return_laplacian=True)
print("energy nan", np.isnan(h).sum(), np.isnan(ps).sum(), np.isnan(lap).sum())
w = np.ones(37, np.float32)
g = m.logpdf_vjp(x, w).cpu().numpy()
print("logpdf_vjp nan", np.isnan(g).sum(), "of", g.size)
g = m.psi_vjp(x, w, 0*w).cpu().numpy()
print("psi_vjp (psi only) nan", np.isnan(g).sum())
g = m.psi_vjp(x, 0*w, w).cpu().numpy()
print("psi_vjp (lap only) nan", np.isnan(g).sum())
idx = np.where(np.isnan(g))[0]
print(idx[:10], idx[-10:] if idx.size else None)
