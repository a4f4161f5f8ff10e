import torch, time, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from waveflow_amd import model_factory
def run(D, knots, B, kernel):
    init = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=knots, n_i_internal_knots=knots,
                                            i_spline_reg=0.05, n_flow_layers=3, box_size=10.0)
    params, psi, log_pdf, _ = init(0, D)
    m = log_pdf.model; m.ensure_params(params); m.set_kernel(kernel)
    g = torch.Generator().manual_seed(1234)
    x = torch.sort((torch.rand(B, D, generator=g) * 2 - 1) * 10.0, dim=-1).values.cuda()
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t = time.perf_counter(); m.log_pdf(x); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(D, knots, kernel, ["%.2f" % t for t in ts])
run(2, 23, 1 << 20, "mfma"); run(2, 23, 1 << 20, "auto"); run(2, 23, 1 << 20, "mfma")
