"""A/B of the taped second-order sweeps: D samples per walker in R3 (WF_GRAD_R3=1) against one sample in RF (default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory

def T(f, n=4):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

for D in [int(a) for a in sys.argv[1:]] or (2, 3, 4, 5, 6, 8):
    g = np.random.default_rng(1234)
    protons = np.linspace(-3, 3, D)
    B = 1 << 14
    x = torch.as_tensor(np.sort(g.uniform(-10, 10, size=(B, D)), -1).astype(np.float32)).cuda()
    res = {}
    for tag in ("R3", "RF"):
        if tag == "R3": os.environ["WF_GRAD_R3"] = "1"
        else: os.environ.pop("WF_GRAD_R3", None)
        init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                                    i_spline_reg=0.05, n_flow_layers=3, box_size=10.0, xu_coord_type="mean")
        params, psi, log_pdf, sample = init_fun(42, D)
        m = psi.model; m.ensure_params(params)
        s, gr = m.vqmc_loss_grad(x, protons, -1.0)
        res[tag] = (gr.double().cpu().numpy(), s.cpu().numpy(), T(lambda: m.vqmc_loss_grad(x, protons, -1.0)))
    a, b = res["R3"], res["RF"]
    print(f"D={D}: loss+grad R3 {B/a[2]:.3e}/s  RF {B/b[2]:.3e}/s  speed-up {a[2]/b[2]:.2f}   rel l2 grad {np.linalg.norm(a[0]-b[0])/np.linalg.norm(a[0]):.2e}  sums {a[1][:2]} {b[1][:2]}")
