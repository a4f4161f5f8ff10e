"""Where is the corrupted value?  flow (u, logdet) vs psi vs log_pdf of the unfenced build, per failing walker."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sorted_walkers  # noqa: E402
from waveflow_amd import checkpoint, model_factory  # noqa: E402

flat = np.load(os.path.join(ROOT, "tests", "golden", "he_checkpoint.npz"))["flat"]
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                            i_spline_reg=0.05, n_flow_layers=3, box_size=10)
params, psi, log_pdf, _ = init_fun(0, 2)
params = checkpoint.unflatten_like(params, flat)
m = log_pdf.model
m.ensure_params(params)
B = 1 << 20
x = torch.from_numpy(sorted_walkers(B, 2, 10.0, 99)).cuda()
m.set_kernel("scalar")
ref_lp = m.log_pdf(x)
ref_psi = m.psi(x)
ref_u, ref_ld = m.flow(x)
m.set_kernel("mfma")
os.environ["WF_MFMA_WAVES"] = os.environ.get("WAVES", "16")
np.set_printoptions(linewidth=200, precision=5)
for r in range(int(os.environ.get("REPS", "10"))):
    u, ld = m.flow(x)
    lp = m.log_pdf(x)
    ps = m.psi(x)
    torch.cuda.synchronize()
    for name, got, ref, tol in (("flow.logdet", ld, ref_ld, 0.02), ("log_pdf", lp, ref_lp, 0.05), ("psi", ps, ref_psi, 1e-3 * float(ref_psi.abs().max()))):
        bad = ((got - ref).abs() > tol) | ~torch.isfinite(got)
        n = int(bad.sum())
        if n:
            w = torch.nonzero(bad).flatten()[:12]
            print(f"launch {r} {name}: {n} bad; walkers {w.cpu().numpy()} (lane {(w % 32).cpu().numpy()})")
            print("     got", got[w].cpu().numpy())
            print("     ref", ref[w].cpu().numpy())
print("done")
