"""A/B of the two local-energy sweeps: D passes in R3 (WF_ENERGY_R3=1) against one forward-Laplacian pass (RF)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory

def T(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

for D in [int(a) for a in sys.argv[1:]] or (2, 3, 4, 5, 6, 8):
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                                i_spline_reg=0.05, n_flow_layers=3, box_size=10.0, xu_coord_type="mean")
    params, psi, log_pdf, sample = init_fun(42, D)
    m = psi.model; m.ensure_params(params)
    g = np.random.default_rng(1234)
    protons = np.linspace(-3, 3, D)
    B = 1 << 15
    x = torch.as_tensor(np.sort(g.uniform(-10, 10, size=(B, D)), -1).astype(np.float32)).cuda()
    res = {}
    for tag in ("R3", "RF"):
        if tag == "R3": os.environ["WF_ENERGY_R3"] = "1"
        else: os.environ.pop("WF_ENERGY_R3", None)
        h, ps, lap = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
        res[tag] = (h.double().cpu().numpy(), ps.double().cpu().numpy(), lap.double().cpu().numpy(), T(lambda: m.hamiltonian(x, protons)))
    a, b = res["R3"], res["RF"]
    sc = np.abs(a[2]).max()
    print(f"D={D}: R3 {B/a[3]:.3e}/s  RF {B/b[3]:.3e}/s  speed-up {a[3]/b[3]:.2f}   psi equal {np.array_equal(a[1], b[1])}  "
          f"max|dlap|/max|lap| {np.abs(a[2]-b[2]).max()/sc:.2e}  rel l2 lap {np.linalg.norm(a[2]-b[2])/np.linalg.norm(a[2]):.2e}  "
          f"rel l2 Hpsi {np.linalg.norm(a[0]-b[0])/np.linalg.norm(a[0]):.2e}")
