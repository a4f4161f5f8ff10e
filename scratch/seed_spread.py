"""Run-to-run spread of the He training curve (batch 256, lr 1e-4, |psi|^2 sampler): medians over 500 steps at 10k / 20k / 30k."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import vqmc
rows = []
for seed in range(8):
    t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=30500, batch_size=256, log_every=10**9)
    t.seed = seed; t.save_dir = f'/tmp/wf_spread_{seed}'; t.exact_sampler = True
    p_, loss = t.start_training(verbose=False)
    l = np.asarray(loss[1:], dtype=float)
    rows.append([np.median(l[a:a + 500]) for a in (5000, 10000, 20000, 30000)])
    print(seed, ' '.join(f'{v:.3f}' for v in rows[-1]), flush=True)
r = np.asarray(rows)
print("min", r.min(0).round(3), "median", np.median(r, 0).round(3), "max", r.max(0).round(3))
