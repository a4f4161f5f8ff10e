import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.chdir(sys.path[0])
from waveflow_amd import checkpoint, model_factory
from waveflow_amd.utils import physics
from oracle import energy_torch as et
flat = np.load('tests/golden/he_checkpoint.npz')['flat']
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10)
params, psi, log_pdf, sample = init_fun(0, 2)
params = checkpoint.unflatten_like(params, flat)
m = psi.model; m.ensure_params(params)
protons, _ = physics.system_catalogue[1]["He"]
g = np.random.default_rng(0)
x = np.sort(g.uniform(-8, 8, size=(128, 2)), -1).astype(np.float32)
wp, wl = g.normal(size=128).astype(np.float32), g.normal(size=128).astype(np.float32)
got = m.psi_vjp(x, wp, wl).cpu().numpy().astype(np.float64)
want = et.psi_vjp(et.he_model(torch.float64), flat, x.astype(np.float64), wp, wl)
w32 = et.psi_vjp(et.he_model(torch.float32), flat, x, wp, wl)
print('rel l2 hip vs f64', np.linalg.norm(got-want)/np.linalg.norm(want), ' torch-f32 vs f64', np.linalg.norm(w32-want)/np.linalg.norm(want))
for B in (256, 4096, 32768, 131072):
    x = torch.as_tensor(np.sort(g.uniform(-8, 8, size=(B, 2)), -1).astype(np.float32)).cuda()
    for _ in range(2):
        m.vqmc_loss_grad(x, protons.reshape(-1), -2.5)
    torch.cuda.synchronize(); t = time.time()
    n = 5
    for _ in range(n):
        s, gr = m.vqmc_loss_grad(x, protons.reshape(-1), -2.5)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    print(f'B={B}: loss+grad {dt*1e3:.2f} ms  -> {B/dt:.3e} walkers/s')
