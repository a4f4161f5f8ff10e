"""The 33-knot ("32-bin") He variant (39 / 38 bases per dimension, 64-row layout of the wave kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory, vqmc
def T(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=33, n_i_internal_knots=33,
                                            i_spline_reg=0.05, n_flow_layers=3, box_size=10.0)
params, psi, log_pdf, sample = init_fun(42, 2)
m = psi.model; m.ensure_params(params)
g = np.random.default_rng(1)
protons = np.zeros(2)
for B in (1 << 14, 1 << 17):
    x = torch.as_tensor(np.sort(g.uniform(-10, 10, size=(B, 2)), -1).astype(np.float32)).cuda()
    print(f"B={B}: psi {B/T(lambda: m.psi(x)):.3e}/s  H psi {B/T(lambda: m.hamiltonian(x, protons)):.3e}/s  loss+grad {B/T(lambda: m.vqmc_loss_grad(x, protons, 0.0)):.3e}/s  sample {B/T(lambda: m.sample(1, B, exact=True)):.3e}/s")
t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=20000, batch_size=256, log_every=10**9)
t.num_knots = 33; t.save_dir = '/tmp/wf_33'; t.exact_sampler = True
t0 = time.time(); p_, loss = t.start_training(verbose=False); dt = time.time() - t0
l = np.asarray(loss[1:])
print(f"training, batch 256: 20000 steps in {dt:.1f} s ({dt/20000*1e3:.3f} ms/step); last 2000 median {np.median(l[-2000:]):.4f}")
