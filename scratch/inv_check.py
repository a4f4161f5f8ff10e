import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from waveflow_amd import checkpoint, model_factory
flat = np.load('tests/golden/he_checkpoint.npz')['flat']
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6, n_flow_layers=3, box_size=10)
params, psi, log_pdf, sample = init_fun(0, 2)
m = psi.model; m.set_params(flat)
om = oracle.he_model(10.0)
g = np.random.default_rng(0)
u = g.uniform(0.001, 0.999, size=(20000, 2)).astype(np.float32)
for exact in (True, False):
    xw = m.inverse(u, exact=exact)                    # wave kernel (B <= 32768)
    big = np.concatenate([u, u[:20000]])               # > 32768 rows -> one-lane-per-walker kernel
    xs = m.inverse(big, exact=exact)[:20000]
    xo = om.inverse(flat, u, exact=exact)
    for nm, a, b in (("wave vs scalar", xw, xs), ("wave vs oracle", xw, xo), ("scalar vs oracle", xs, xo)):
        d = np.abs(a - b)
        print(f"exact={exact} {nm}: identical {np.mean(d == 0):.3f}  median {np.median(d):.2e}  p99 {np.quantile(d, 0.99):.2e}  max {d.max():.2e}")
