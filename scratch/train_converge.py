"""End to end through the matrix-core paths: VQMC training of the He box at a large batch (staged sampler, H psi, tile gradient, Adam; one hipGraph replay per
step) -- does <E_L> approach the exact ground-state energy of this Hamiltonian (-1.8161, finite-difference diagonalisation: DESIGN 4.5)?
usage: [B=32768] [EPOCHS=4000] [LR=1e-3] python3 scratch/train_converge.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from waveflow_amd import vqmc
B, E, LR = int(os.environ.get("B", 32768)), int(os.environ.get("EPOCHS", 4000)), float(os.environ.get("LR", 1e-3))
ALL = {"all": ("matrix-core paths", {}), "wave": ("wave kernels", {"WF_GRAD_TILE_MIN": "0", "WF_SAMPLE_TILE_MIN": "0", "WF_ENERGY_TILE_MIN": "0"}),
       "sampler": ("staged sampler only", {"WF_GRAD_TILE_MIN": "0", "WF_ENERGY_TILE_MIN": "0"}), "grad": ("tile gradient / H psi only", {"WF_SAMPLE_TILE_MIN": "0"})}
for tag, env in (ALL[k] for k in os.environ.get("CONFIGS", "all wave").split()):
    os.environ.update(env)
    t = vqmc.ModelTrainer(system_name="He", learning_rate=LR, box_length=10, num_epochs=E, batch_size=B, log_every=10 ** 9)
    t.save_dir = f"/tmp/wf_converge_{abs(hash(tag)) % 1000}"
    t.exact_sampler = True
    t.seed = int(os.environ.get("SEED", 2))
    t0 = time.perf_counter()
    params, loss = t.start_training(verbose=False)
    dt = time.perf_counter() - t0
    l = np.asarray(loss[1:], dtype=np.float64)
    print(f"{tag}: {E} steps of {B} walkers in {dt:.2f} s; <E_L> first 10 steps {l[:10].mean():+.3f}, steps {E // 2 - 100}..{E // 2} {l[E // 2 - 100:E // 2].mean():+.4f}, "
          f"last 200 {l[-200:].mean():+.5f} +- {l[-200:].std() / np.sqrt(200):.5f} (exact -1.8161); finite {np.isfinite(l).all()}", flush=True)
    tail = l[-2000:]
    print(f"    last 2000 steps: median {np.median(tail):+.5f}, 1 % / 99 % quantiles {np.quantile(tail, 0.01):+.4f} / {np.quantile(tail, 0.99):+.4f}, min {tail.min():+.3f} max {tail.max():+.3f}, "
          f"steps beyond 0.1 of the median: {(np.abs(tail - np.median(tail)) > 0.1).sum()}", flush=True)
    for k in env:
        del os.environ[k]
