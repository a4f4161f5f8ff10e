import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from waveflow_amd import checkpoint, model_factory
flat = np.load("tests/golden/he_checkpoint.npz")["flat"]
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10)
params, psi, log_pdf, _ = init_fun(0, 2)
m = log_pdf.model; m.set_params(flat)
x = np.sort(np.random.default_rng(1234).uniform(-10, 10, size=(20000, 2)).astype(np.float32), -1)
lp, u = m.log_pdf(x, return_sample=True)
uf, ld = m.flow(x)
ps = m.psi(x)
np.savez("gpurun_out/dbg1.npz", lp=lp, u=u, uf=uf, ld=ld, ps=ps)
