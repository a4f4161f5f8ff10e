"""k_mfma launch time against the workgroup shape (WF_MFMA_WAVES, read at every launch): He 23 knots (NBK = 1) and 33 knots (NBK = 2), 2^20 walkers."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import sorted_walkers
from waveflow_amd import model_factory
x = torch.from_numpy(sorted_walkers(1 << 20, 2, 10.0, 99)).cuda()
for knots in (23, 33):
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=knots, n_i_internal_knots=knots,
                                                i_spline_reg=0.05, n_flow_layers=3, box_size=10)
    params, psi, log_pdf, _ = init_fun(0, 2)
    m = log_pdf.model; m.ensure_params(params); m.set_kernel("mfma")
    for rep in range(2):
        for w in (8, 12, 16):
            os.environ["WF_MFMA_WAVES"] = str(w)
            for _ in range(150): m.log_pdf(x)
            ts = []
            for _ in range(50):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); m.log_pdf(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            print(f"knots {knots} waves {w}: median {np.median(ts):.4f} ms  min {np.min(ts):.4f}")
