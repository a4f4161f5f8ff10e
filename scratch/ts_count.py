"""Diagnostics (-DWF_TS_COUNT build): histogram of the number of proposals the staged sampler needs for the prior's second column."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("auto")
x, lat = m.sample(7, 1 << 17, return_latent=True, exact=True)
n = lat[:, 1].cpu().numpy()
print("proposals per walker: mean %.2f median %.0f p99 %.0f max %.0f ; acceptance %.3f" % (n.mean(), np.median(n), np.quantile(n, 0.99), n.max(), 1.0 / n.mean()))
w = n.reshape(-1, 64).max(1)
print("per wave (64 consecutive walkers): mean of the maximum %.1f" % w.mean())
