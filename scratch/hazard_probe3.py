"""Debug taps of the Waveflow prior (WF_DBG_PRIOR builds write two intermediates in place of u): reference = the fenced build,
test = an unfenced build; both run the same arithmetic, so every difference is corruption."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sorted_walkers  # noqa: E402
from waveflow_amd import checkpoint, model_factory  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
flat = np.load(os.path.join(ROOT, "tests", "golden", "he_checkpoint.npz"))["flat"]
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                            i_spline_reg=0.05, n_flow_layers=3, box_size=10)
params, psi, log_pdf, _ = init_fun(0, 2)
params = checkpoint.unflatten_like(params, flat)
m = log_pdf.model
m.ensure_params(params)
B = 1 << 20
x = torch.from_numpy(sorted_walkers(B, 2, 10.0, 99)).cuda()
m.set_kernel("mfma")
np.set_printoptions(linewidth=220, precision=7)
if mode == "ref":
    os.environ["WF_MFMA_WAVES"] = "8"
    lp, u = m.log_pdf(x, return_sample=True)
    torch.save({"lp": lp.cpu(), "u": u.cpu()}, path)
    print("saved", path)
else:
    ref = torch.load(path)
    rlp, ru = ref["lp"].cuda(), ref["u"].cuda()
    for waves in ("8", "12", "16"):
        os.environ["WF_MFMA_WAVES"] = waves
        n_bad = n_a = n_b = 0
        shown = 0
        for r in range(int(os.environ.get("REPS", "30"))):
            lp, u = m.log_pdf(x, return_sample=True)
            torch.cuda.synchronize()
            bad = (lp != rlp)
            da, db = (u[:, 0] != ru[:, 0]), (u[:, 1] != ru[:, 1])
            n_bad += int(bad.sum()); n_a += int(da.sum()); n_b += int(db.sum())
            anyb = bad | da | db
            if int(anyb.sum()) and shown < 3:
                shown += 1
                w = torch.nonzero(anyb).flatten()[:8]
                print(f"  waves {waves} launch {r}: walkers {w.cpu().numpy()} lanes {(w % 32).cpu().numpy()}")
                print("     lp  got", lp[w].cpu().numpy(), "\n         ref", rlp[w].cpu().numpy())
                print("     A   got", u[w, 0].cpu().numpy(), "\n         ref", ru[w, 0].cpu().numpy())
                print("     B   got", u[w, 1].cpu().numpy(), "\n         ref", ru[w, 1].cpu().numpy())
        print(f"{os.path.basename(os.environ.get('WF_LIB', ''))} waves {waves}: logp differs {n_bad}, tap A differs {n_a}, tap B differs {n_b}", flush=True)
