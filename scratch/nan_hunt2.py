"""Eager wf_vqmc_train_step until the loss ring shows a non-finite entry; then look at the step's walkers in the workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import vqmc
from waveflow_amd.utils import physics
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
psi, log_pdf, sample, st0, opt_update, get_params = vqmc.create_train_state(10, 1e-4, 2, rng=0)
protons = physics.system_catalogue[1]['He'][0].reshape(-1)
m = psi.model
st = m.make_train_state(st0.x, st0.m, st0.v, 1, ring_len=128)
m.set_params_device(st0.x)
al = lambda n: (n + 255) // 256 * 256
for step in range(1, 600):
    prev = st0.x.clone()
    m.train_step(st, 12345, B, protons, 1e-4, exact_sampler=True)
    r = st["ring"].cpu().numpy()[step % 128]
    if step % 100 == 0:
        rr = st["ring"].cpu().numpy()
        st["running_average"].fill_(float(np.mean([rr[e % 128, 0] / rr[e % 128, 2] for e in range(step - 99, step)])))
    if not np.isfinite(r).all() or not torch.isfinite(st0.x).all():
        ws = st["ws"]
        x = ws[:B * 2 * 4].view(torch.float32).view(B, 2).clone()
        el = ws[al(B * 2 * 4):al(B * 2 * 4) + B * 4].view(torch.float32).clone()
        g = ws[al(B * 2 * 4) + al(B * 4):al(B * 2 * 4) + al(B * 4) + m.n_params * 4].view(torch.float32).clone()
        bad = torch.nonzero(~torch.isfinite(el)).flatten().tolist()
        print("step", step, "ring", r, "bad e_loc walkers", bad[:8], "x", x[bad[:8]].cpu().numpy() if bad else "", "bad grad entries", int((~torch.isfinite(g)).sum()),
              "params finite", bool(torch.isfinite(st0.x).all()), "prev params finite", bool(torch.isfinite(prev).all()))
        np.savez("gpurun_out/nan_case.npz", x=x.cpu().numpy(), flat=prev.cpu().numpy(), el=el.cpu().numpy())
        m.set_params_device(prev)
        h, ps, lap = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
        print("same walkers, wf_hamiltonian_fwd: non-finite", int((~torch.isfinite(h)).sum()), "psi==0:", int((ps == 0).sum()), "min |psi|", float(ps.abs().min()))
        if bad:
            i = bad[0]
            print("walker", i, x[i].cpu().numpy(), "Hpsi", float(h[i]), "psi", float(ps[i]), "lap", float(lap[i]))
        break
else:
    print("no NaN")
