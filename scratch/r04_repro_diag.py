"""bitwise reproducibility of the matrix-core gradient path: N repeats of psi_vjp, which leaves differ from the first result"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from waveflow_amd import checkpoint, model_factory
from test_gpu_grad import sorted_walkers, _leaves
for kn in (33, 23):
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=kn, n_i_internal_knots=kn,
                                                i_spline_reg=0.05, n_flow_layers=3, box_size=10.0)
    params, psi, log_pdf, sample = init_fun(3, 2)
    m = psi.model
    m.ensure_params(params)
    B = int(os.environ.get("B", 50001))
    g = np.random.default_rng(8)
    xb = torch.as_tensor(sorted_walkers(B, 2, 9.5, 7)).cuda()
    wb1 = torch.as_tensor(g.normal(size=B).astype(np.float32)).cuda()
    wb2 = torch.as_tensor((0.1 * g.normal(size=B)).astype(np.float32)).cuda()
    first = m.psi_vjp(xb, wb1, wb2).cpu().numpy()
    nbad = 0
    for rep in range(int(os.environ.get("REPS", 30))):
        cur = m.psi_vjp(xb, wb1, wb2).cpu().numpy()
        if not np.array_equal(cur, first):
            nbad += 1
            if nbad <= 3:
                for i, (a, b) in enumerate(zip(_leaves(checkpoint.unflatten_like(params, cur)), _leaves(checkpoint.unflatten_like(params, first)))):
                    d = (a != b)
                    if d.any():
                        idx = np.argwhere(d)
                        print(f"  knots {kn} rep {rep}: leaf {i} {a.shape}: {d.sum()} entries differ, first at {idx[0]}, rows {np.unique(idx[:, 0])[:8]} cols {np.unique(idx[:, -1])[:12]}, max rel {np.abs(a - b)[d].max() / np.abs(b).max():.2e}")
    print(f"knots {kn}: {nbad} of {os.environ.get('REPS', 30)} repeats differ from the first")
