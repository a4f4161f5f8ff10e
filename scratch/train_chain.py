"""Captured-step training of D-electron chains (no reference system exists beyond 2 electrons: config C4): finiteness and descent."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory
from waveflow_amd.core import flatten_params
for D in (3, 4, 6):   # (D = 8 from a random start diverges with these settings: E_L of 1e30 by step 270 -- optimisation, not arithmetic)
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                                i_spline_reg=0.05, n_flow_layers=3, box_size=12.0, xu_coord_type=os.environ.get("XU", "first"))
    params, psi, log_pdf, sample = init_fun(1, D)
    m = psi.model
    x = torch.as_tensor(flatten_params(params).astype(np.float32)).cuda()
    st = m.make_train_state(x, torch.zeros_like(x), torch.zeros_like(x), 1, ring_len=4096, defer_eval_tables=True)
    m.set_params_device(x)
    protons = np.linspace(-(D - 1), D - 1, D).astype(np.float32)      # one proton per electron, spacing 2
    steps, batch = 3000, 512
    m.train_step(st, 5, batch, protons, 2e-4, exact_sampler=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps - 1):
        m.train_step(st, 5, batch, protons, 2e-4, exact_sampler=True)
        if (i + 2) % 100 == 0:
            r = st["ring"].cpu().numpy()
            idx = [(e % 4096) for e in range(i + 2 - 99, i + 2)]
            st["running_average"].fill_(float(np.mean(r[idx, 0] / r[idx, 2])))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = st["ring"].cpu().numpy()[1:steps + 1]
    l = r[:, 0] / r[:, 2]
    print(f"D={D}: {steps} steps of {batch} walkers, {dt/steps*1e3:.3f} ms/step (eager); finite {bool(np.isfinite(l).all() and torch.isfinite(x).all())}; "
          f"median <E_L> first 200: {np.median(l[:200]):.3f}  last 200: {np.median(l[-200:]):.3f}")
