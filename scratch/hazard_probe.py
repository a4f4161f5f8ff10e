"""Runs the He-shape MFMA kernel of the library in $WF_LIB at 2^20 walkers, repeatedly, for 8 / 12 / 16 waves per workgroup, and
reports where results differ from launch to launch and from the scalar kernel."""
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
ref, ref_u = m.log_pdf(x, return_sample=True)
m.set_kernel("mfma")
reps = int(os.environ.get("REPS", "12"))
tag = os.path.basename(os.environ.get("WF_LIB", "default"))
for cfg in os.environ.get("CONFIGS", "8x1 12x1 16x1 8x2 4x2").split():
    waves, tiles = cfg.split("x")
    os.environ["WF_MFMA_WAVES"] = waves
    os.environ["WF_MFMA_TILES"] = tiles
    bad_total, unstable, first = 0, 0, None
    report = []
    for r in range(reps):
        lp, u = m.log_pdf(x, return_sample=True)
        torch.cuda.synchronize()
        if first is None:
            first = lp.clone()
        unstable += int((lp != first).sum().item())
        bad = ((lp - ref).abs() > 0.05) | ((u - ref_u).abs() > 1e-3).any(-1) | ~torch.isfinite(lp)
        n = int(bad.sum().item())
        bad_total += n
        if n and len(report) < 3:
            w = torch.nonzero(bad).flatten().cpu().numpy()
            tiles = np.unique(w // 32)
            du = (u - ref_u).abs()[bad].max(0).values.cpu().numpy()
            report.append(f"    launch {r}: {n} walkers in {len(tiles)} tiles; lanes-in-tile {sorted(set((w % 32).tolist()))[:40]}; "
                          f"max|du| {du}; tiles {tiles[:6].tolist()}")
    print(f"{tag:28s} waves x tiles {cfg:>5s}: bad walker-launches {bad_total:6d}  launch-to-launch differing values {unstable:6d}", flush=True)
    for line in report:
        print(line, flush=True)
