"""A/B timing of several builds in ONE run: every library in its own child process is too noisy (clock state), so this script takes
a list of libraries, and for ROUNDS rounds runs each library's timing child back to back; reports per library the median of the
per-round medians and the overall minimum.   usage: time_ab.py lib1 lib2 ... (names under scratch/variants, or 'default')"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import sorted_walkers
from waveflow_amd import checkpoint, model_factory
flat = np.load(os.path.join(%r, "tests", "golden", "he_checkpoint.npz"))["flat"]
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23,
                                            i_spline_reg=0.05, n_flow_layers=3, box_size=10)
params, psi, log_pdf, _ = init_fun(0, 2)
params = checkpoint.unflatten_like(params, flat)
m = log_pdf.model; m.ensure_params(params); m.set_kernel("mfma")
x = torch.from_numpy(sorted_walkers(1 << 20, 2, 10.0, 99)).cuda()
for cfg in os.environ["CONFIGS"].split():
    w, t = cfg.split("x"); os.environ["WF_MFMA_WAVES"] = w; os.environ["WF_MFMA_TILES"] = t
    for _ in range(10): m.log_pdf(x)
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.log_pdf(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(cfg, np.median(ts), np.min(ts))
''' % (ROOT, ROOT, ROOT)

libs = sys.argv[1:]
rounds = int(os.environ.get("ROUNDS", "3"))
os.environ.setdefault("CONFIGS", "16x1")
res = {}
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["WF_LIB"] = os.path.join(ROOT, "scratch", "variants", f"libwf_{lib}.so"); env["WF_LIB_EXPERIMENT"] = "1"
        else:
            env.pop("WF_LIB", None)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            p = line.split()
            if len(p) == 3:
                res.setdefault((lib, p[0]), []).append((float(p[1]), float(p[2])))
for (lib, cfg), v in res.items():
    med = np.median([a for a, _ in v]); mn = min(b for _, b in v)
    print(f"{lib:16s} {cfg:>5s}: median of round medians {med:.4f} ms   min {mn:.4f} ms   rounds {['%.4f' % a for a, _ in v]}")
