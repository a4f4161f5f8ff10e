import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory
from waveflow_amd.core import flatten_params
D = 8
init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                            i_spline_reg=0.05, n_flow_layers=3, box_size=12.0, xu_coord_type="first")
params, psi, log_pdf, sample = init_fun(1, D)
m = psi.model
x = torch.as_tensor(flatten_params(params).astype(np.float32)).cuda()
st = m.make_train_state(x, torch.zeros_like(x), torch.zeros_like(x), 1, ring_len=4096)
m.set_params_device(x)
protons = np.linspace(-(D - 1), D - 1, D).astype(np.float32)
B = 512
al = lambda n: (n + 255) // 256 * 256
for step in range(1, 3000):
    prev = x.clone()
    m.train_step(st, 5, B, protons, 2e-4, exact_sampler=True)
    r = st["ring"].cpu().numpy()[step % 4096]
    if step % 100 == 0:
        rr = st["ring"].cpu().numpy(); idx = [(e % 4096) for e in range(step - 99, step)]
        st["running_average"].fill_(float(np.mean(rr[idx, 0] / rr[idx, 2])))
    if not np.isfinite(r).all() or not torch.isfinite(x).all():
        ws = st["ws"]
        xs = ws[:B * D * 4].view(torch.float32).view(B, D).clone()
        el = ws[al(B * D * 4):al(B * D * 4) + B * 4].view(torch.float32).clone()
        bad = torch.nonzero(~torch.isfinite(el)).flatten().tolist()
        print("step", step, "ring", r, "params finite", bool(torch.isfinite(x).all()), "prev finite", bool(torch.isfinite(prev).all()), "bad e_loc", bad[:5],
              "|el| max finite", float(el[torch.isfinite(el)].abs().max()), "running avg", float(st["running_average"]))
        m.set_params_device(prev)
        h, ps, lap = m.hamiltonian(xs, protons, return_psi=True, return_laplacian=True)
        os.environ["WF_ENERGY_R3"] = "1"; h3, ps3, lap3 = m.hamiltonian(xs, protons, return_psi=True, return_laplacian=True)
        print("hamiltonian on the step's walkers with the previous parameters: non-finite RF", int((~torch.isfinite(h)).sum()), "R3", int((~torch.isfinite(h3)).sum()),
              "min |psi|", float(ps.abs().min()), "max |lap|", float(lap[torch.isfinite(lap)].abs().max()))
        for i in bad[:3]:
            print(" walker", i, xs[i].cpu().numpy(), "psi", float(ps[i]), "lap RF", float(lap[i]), "lap R3", float(lap3[i]), "H RF", float(h[i]))
        s, g = m.vqmc_loss_grad(xs, protons, float(st["running_average"]))
        print("loss_grad on them: sums", s.cpu().numpy(), "grad non-finite", int((~torch.isfinite(g)).sum()))
        break
else:
    print("no NaN")
