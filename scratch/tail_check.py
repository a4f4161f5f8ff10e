import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import checkpoint, model_factory
flat = np.load('tests/golden/he_checkpoint.npz')['flat']
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6, n_flow_layers=3, box_size=10)
params, psi, log_pdf, sample = init_fun(0, 2)
m = psi.model; m.set_params(flat)
for tag, nper, reps in (("wave", 32768, 40), ("one-lane", 65536, 20)):
    cnt = np.zeros((2, 6)); tot = 0
    for s in range(reps):
        x, lat = m.sample(777 + s, nper, return_latent=True, exact=True)
        l = lat.cpu().numpy(); tot += nper
        for c in range(2):
            cnt[c] += [(l[:, c] > 0.99).sum(), (l[:, c] > 0.98).sum(), (l[:, c] > 0.95).sum(), (l[:, c] < 0.01).sum(), (l[:, c] < 0.02).sum(), (l[:, c] < 0.05).sum()]
    print(tag, tot, "col0 [>.99 >.98 >.95 <.01 <.02 <.05]:", (cnt[0] / tot * 1e5).round(1), "col1:", (cnt[1] / tot * 1e5).round(1), "(per 1e5)")
