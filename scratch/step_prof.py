"""Whole VQMC training steps at a large batch (sampler + loss + gradient + Adam + image refresh, one hipGraph replay each) with the gradient on the
matrix cores and on the wave sweeps (WF_GRAD_TILE_MIN=0), and the sampler alone.   usage: [B=131072] python3 scratch/step_prof.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd import vqmc

B = int(os.environ.get("B", 1 << 17))
m, flat = bench.he_model("auto")
for exact in (False, True):
    for n in (B, 1 << 20):
        m.sample(3, n, exact=exact); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            m.sample(4 + i, n, exact=exact)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"sample exact={exact} n={n}: {dt * 1e3:.3f} ms = {n / dt:.3e} walkers/s", flush=True)
for tm in (("16384", "0") if B <= (1 << 17) else ("16384",)):
    os.environ["WF_GRAD_TILE_MIN"] = tm
    tr = vqmc.ModelTrainer(system_name="He", learning_rate=1e-4, box_length=10, num_epochs=int(os.environ.get("EPOCHS", 60)), batch_size=B, log_every=10 ** 9)
    tr.save_dir = "/tmp/wf_step_prof"
    tr.exact_sampler = bool(int(os.environ.get("EXACT", "1")))
    t0 = time.perf_counter()
    params, loss = tr.start_training(verbose=False)
    dt = (time.perf_counter() - t0)
    l = np.asarray(loss[1:], dtype=np.float64)
    print(f"WF_GRAD_TILE_MIN={tm}: {tr.num_epochs} steps of {B} walkers in {dt:.3f} s (incl. setup) ; last losses {l[-3:]} finite {np.isfinite(l).all()}", flush=True)
