import torch, time, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from waveflow_amd import model_factory
def run(D, knots, B, layers=3, k=6):
    init = model_factory.get_waveflow_model(D, base_spline_degree=k, i_spline_degree=k, n_prior_internal_knots=knots, n_i_internal_knots=knots,
                                            i_spline_reg=0.05, n_flow_layers=layers, box_size=10.0)
    params, psi, log_pdf, _ = init(0, D)
    m = log_pdf.model; m.ensure_params(params)
    g = torch.Generator().manual_seed(1234)
    x = torch.sort((torch.rand(B, D, generator=g) * 2 - 1) * 10.0, dim=-1).values.cuda()
    for _ in range(2): m.log_pdf(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 5
    for _ in range(n): m.log_pdf(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"D={D} knots={knots} (I bases {m.i_nb}) B={B}: {dt*1e3:.2f} ms  {B/dt:.3e} evals/s")
run(2, 23, 1 << 20); run(2, 33, 1 << 20); run(8, 23, 1 << 18); run(4, 23, 1 << 18); run(3, 23, 1 << 18)
