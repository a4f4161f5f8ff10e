"""Large random sweep of the derivative kernels: no non-finite local energies / gradients; RF and R3 sweeps agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import checkpoint, model_factory, vqmc
flat_ck = np.load('tests/golden/he_checkpoint.npz')['flat']
def he(knots):
    f = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=knots, n_i_internal_knots=knots,
                                         i_spline_reg=0.05, n_flow_layers=3, box_size=10)
    return f(0, 2)
protons = np.zeros(2)
cases = []
params, psi, log_pdf, sample = he(23)
cases.append(("shipped checkpoint", psi, checkpoint.unflatten_like(params, flat_ck)))
cases.append(("seeded init", psi, params))
t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=30000, batch_size=256, log_every=10**9)
t.save_dir = '/tmp/wf_sweep'; t.exact_sampler = True
p_tr, _ = t.start_training(verbose=False)
cases.append(("trained 30k steps", t.psi, p_tr))
p33, psi33, _, _ = he(33)
cases.append(("33 knots, seeded init", psi33, p33))
g = np.random.default_rng(5)
for name, ps, prm in cases:
    m = ps.model; m.ensure_params(prm)
    bad_h = bad_g = 0; n = 0; worst = 0.0
    for rep in range(8):
        B = 1 << 19
        x = torch.as_tensor(np.sort(g.uniform(-10, 10, size=(B, 2)), -1).astype(np.float32)).cuda()
        h, p, lap = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
        bad_h += int((~torch.isfinite(h)).sum()); n += B
        os.environ["WF_ENERGY_R3"] = "1"
        h3 = m.hamiltonian(x, protons)
        os.environ.pop("WF_ENERGY_R3")
        ok = torch.isfinite(h) & torch.isfinite(h3)
        worst = max(worst, float(((h - h3).abs() / (h3.abs() + 1e-3 * h3.abs().median()))[ok].quantile(0.999)))
        xs = m.sample(rep, 1 << 16, exact=True)
        s, gr = m.vqmc_loss_grad(xs, protons, -1.8)
        bad_g += int((~torch.isfinite(gr)).sum()) + int(not np.isfinite(s.cpu().numpy()).all())
    print(f"{name}: {n} uniform walkers: non-finite H psi {bad_h}; RF vs R3 relative difference, 99.9 % quantile {worst:.1e}; non-finite loss/gradient entries over 8 x 65536 |psi|^2 walkers: {bad_g}")
