"""Kernel time of the He-shape log_pdf at 2^20 walkers for the library in $WF_LIB, per workgroup size (HIP events, median of 30)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sorted_walkers  # noqa: E402
from waveflow_amd import checkpoint, model_factory  # noqa: E402

knots = int(os.environ.get("KNOTS", "23"))
flat = np.load(os.path.join(ROOT, "tests", "golden", "he_checkpoint.npz"))["flat"]
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=knots, n_i_internal_knots=knots,
                                            i_spline_reg=0.05, n_flow_layers=3, box_size=10)
params, psi, log_pdf, _ = init_fun(0, 2)
if knots == 23:
    params = checkpoint.unflatten_like(params, flat)
m = log_pdf.model
m.ensure_params(params)
B = 1 << 20
x = torch.from_numpy(sorted_walkers(B, 2, 10.0, 99)).cuda()
m.set_kernel("mfma")
tag = os.path.basename(os.environ.get("WF_LIB", "default"))
for cfg in os.environ.get("CONFIGS", "8x1 12x1 16x1 8x2 4x2").split():
    waves, tiles = cfg.split("x")
    os.environ["WF_MFMA_WAVES"] = waves
    os.environ["WF_MFMA_TILES"] = tiles
    for _ in range(5):
        m.log_pdf(x)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.log_pdf(x)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{tag:28s} knots {knots} waves x tiles {cfg:>5s}: median {np.median(ts):.4f} ms  min {np.min(ts):.4f} ms", flush=True)
