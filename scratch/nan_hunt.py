"""Host-stepped training (RF sweeps) until a non-finite local energy or gradient shows; then the same walkers through the R3 sweeps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import vqmc
from waveflow_amd.utils import physics
B = 4096
psi, log_pdf, sample, st, opt_update, get_params = vqmc.create_train_state(10, 1e-4, 2, rng=0)
protons = physics.system_catalogue[1]['He'][0].reshape(-1)
m = psi.model
avg = 0.0
for step in range(1, 400):
    params = get_params(st)
    x = sample(step, params, B, exact_inverse=True)
    m.ensure_params(params)
    sums, grad = m.vqmc_loss_grad(x, protons, avg)
    s = sums.cpu().numpy()
    if not np.isfinite(s).all() or not torch.isfinite(grad).all():
        print("step", step, "sums", s, "grad finite", bool(torch.isfinite(grad).all()), "n bad grad", int((~torch.isfinite(grad)).sum()))
        h, ps, lap = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
        bad = torch.nonzero(~torch.isfinite(h / (ps + 1e-8))).flatten()
        print("walkers with bad E_L from wf_hamiltonian_fwd (RF, untaped):", bad.tolist()[:10])
        # per-walker: which walkers give a non-finite gradient
        xs = x.cpu().numpy()
        flat = st.x.cpu().numpy().copy()
        np.savez("gpurun_out/nan_case.npz", x=xs, flat=flat)
        culprits = []
        for lo in range(0, B, 256):
            _, g = m.vqmc_loss_grad(x[lo:lo + 256].contiguous(), protons, avg)
            if not torch.isfinite(g).all():
                for i in range(lo, lo + 256):
                    s1, g1 = m.vqmc_loss_grad(x[i:i + 1].contiguous(), protons, avg)
                    if not torch.isfinite(g1).all() or not np.isfinite(s1.cpu().numpy()).all():
                        culprits.append(i)
        print("culprit walkers:", culprits, xs[culprits] if culprits else "")
        u = m.log_pdf(x, return_sample=True) if hasattr(m, "log_pdf") else None
        break
    st = opt_update(step, grad, st)
    if step % 100 == 0:
        avg = 0.0
else:
    print("no NaN in 400 steps")
