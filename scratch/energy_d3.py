import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory
os.environ["WF_ENERGY_R3"] = "1"
for D in (3, 4):
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                                i_spline_reg=0.05, n_flow_layers=3, box_size=10.0, xu_coord_type="mean")
    params, psi, log_pdf, sample = init_fun(42, D)
    m = psi.model; m.ensure_params(params)
    g = np.random.default_rng(1234)
    x = torch.as_tensor(np.sort(g.uniform(-10, 10, size=(1 << 15, D)), -1).astype(np.float32)).cuda()
    for _ in range(5): m.hamiltonian(x, np.linspace(-3, 3, D))
    torch.cuda.synchronize()
    for i in range(4):
        t = time.perf_counter(); m.hamiltonian(x, np.linspace(-3, 3, D)); torch.cuda.synchronize(); print(D, "call", i, (time.perf_counter() - t) * 1e3, "ms")
    t = time.perf_counter()
    for i in range(5): m.hamiltonian(x, np.linspace(-3, 3, D))
    torch.cuda.synchronize(); print(D, "5 calls back to back", (time.perf_counter() - t) / 5 * 1e3, "ms each")
