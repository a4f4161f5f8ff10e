import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
m, _ = bench.he_model("auto")
xa, la = m.sample(11, 60000, return_latent=True, exact=True)
os.environ["WF_SAMPLE_ONE_LANE"] = "1"
xo, lo = m.sample(11, 60000, return_latent=True, exact=True)
d = (la - lo).abs()
print("latent col0 equal:", bool((la[:, 0] == lo[:, 0]).all()), " col1 equal fraction:", float((la[:, 1] == lo[:, 1]).float().mean()), " max |diff| col1:", float(d[:, 1].max()))
idx = torch.nonzero(la[:, 1] != lo[:, 1]).flatten()[:10]
print("first differing walkers:", idx.tolist())
print(torch.stack([la[idx, 1], lo[idx, 1]], 1))
