"""Two-sample Kolmogorov-Smirnov of the staged sampler's draws against the one-walker-per-wave kernel's, 2^20 walkers each (latent columns and coordinates)."""
import os, sys
import numpy as np
import torch
from scipy import stats
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("auto")
n = 1 << 20
os.environ["WF_SAMPLE_TILE_MIN"] = "16384"
xa, la = m.sample(101, n, return_latent=True, exact=True)
os.environ["WF_SAMPLE_TILE_MIN"] = "0"
os.environ["WF_WAVE_SAMPLE_MAX"] = "100000000"
xb, lb = m.sample(202, n, return_latent=True, exact=True)
for name, a, b in (("latent0", la[:, 0], lb[:, 0]), ("latent1", la[:, 1], lb[:, 1]), ("x0", xa[:, 0], xb[:, 0]), ("x1", xa[:, 1], xb[:, 1]),
                   ("x1-x0", xa[:, 1] - xa[:, 0], xb[:, 1] - xb[:, 0])):
    r = stats.ks_2samp(a.cpu().numpy(), b.cpu().numpy())
    print(f"{name}: KS statistic {r.statistic:.5f} (1 % level at this n: {1.63 * np.sqrt(2 / n):.5f})  p = {r.pvalue:.3f}")
