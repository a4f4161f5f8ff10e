import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np, torch
from waveflow_amd import model_factory
D = 8
init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10.0, xu_coord_type="mean")
params, psi, log_pdf, sample = init_fun(42, D)
m = psi.model; m.ensure_params(params)
g = np.random.default_rng(1234)
protons = np.linspace(-7, 7, 8)
def T(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for B in (1 << 14, 1 << 16):
    x = torch.as_tensor(np.sort(g.uniform(-10, 10, size=(B, D)), -1).astype(np.float32)).cuda()
    print(f"D=8 B={B}: psi {B/T(lambda: m.psi(x)):.3e}/s  H psi {B/T(lambda: m.hamiltonian(x, protons)):.3e}/s  loss+grad {B/T(lambda: m.vqmc_loss_grad(x, protons, 0.0)):.3e}/s")
x = sample(1, params, 4096, exact_inverse=True)
print("sample 4096 (wave):", T(lambda: sample(1, params, 4096, exact_inverse=True)) * 1e3, "ms")
