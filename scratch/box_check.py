import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import model_factory
for D in (2, 3, 4):
    for nl in (0, 1):
        init_fun = model_factory.get_waveflow_model(D, base_spline_degree=5, i_spline_degree=5, n_prior_internal_knots=16, n_i_internal_knots=16,
                                                    i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6, n_flow_layers=nl, box_size=7.0, xu_coord_type="mean")
        params, psi, log_pdf, sample = init_fun(3, D)
        m = psi.model; m.ensure_params(params)
        x = np.sort(np.random.default_rng(0).uniform(-7, 7, size=(6, D)), -1).astype(np.float32)
        u, ld = m.flow(x)
        x2 = m.inverse(u, exact=True)
        print("D", D, "layers", nl, "u range", u.min(), u.max(), "max |x2 - x|", np.abs(x2 - x).max())
        if np.abs(x2 - x).max() > 1e-2: print(x[:2], u[:2], x2[:2])
