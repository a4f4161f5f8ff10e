import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["WF_POISON"] = "1"
from waveflow_amd import model_factory
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=33, n_i_internal_knots=33, i_spline_reg=0.05,
                                            i_spline_reverse_fun_tol=1e-6, n_flow_layers=2, box_size=10, xu_coord_type="mean")
params, psi, log_pdf, sample = init_fun(3, 2)
m = psi.model
m.ensure_params(params)
for exact in (True, False):
    os.environ["WF_WAVE_SAMPLE_MAX"] = "100000000"
    xa, la = m.sample(7, 30000, return_latent=True, exact=exact)
    xa2, la2 = m.sample(7, 30000, return_latent=True, exact=exact)
    os.environ["WF_WAVE_SAMPLE_MAX"] = "0"
    xb, lb = m.sample(8, 40000, return_latent=True, exact=exact)
    xb2, lb2 = m.sample(8, 40000, return_latent=True, exact=exact)
    print("exact", exact, "wave reproducible", torch.equal(xa, xa2), torch.equal(la, la2), "lane reproducible", torch.equal(xb, xb2), torch.equal(lb, lb2))
    for name, a, b in (("x", xa, xb), ("lat", la, lb)):
        print("  ", name, "wave mean", a.mean(0).tolist(), "std", a.std(0).tolist(), "nan", torch.isnan(a).sum().item())
        print("  ", name, "lane mean", b.mean(0).tolist(), "std", b.std(0).tolist(), "nan", torch.isnan(b).sum().item())
