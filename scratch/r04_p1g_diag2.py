import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
m, _ = bench.he_model("auto")
os.environ.pop("WF_SAMPLE_ONE_LANE", None)
xa, la = m.sample(11, 60000, return_latent=True, exact=True)
os.environ["WF_SAMPLE_ONE_LANE"] = "1"
xo, lo = m.sample(11, 60000, return_latent=True, exact=True)
print(os.environ.get("WF_LIB", "default")[-16:], "debug value equal fraction:", float((la[:, 0] == lo[:, 0]).float().mean()), " max rel diff:", float(((la[:, 0] - lo[:, 0]).abs() / lo[:, 0].abs().clamp_min(1e-30)).max()),
      " col1 equal fraction:", float((la[:, 1] == lo[:, 1]).float().mean()))
