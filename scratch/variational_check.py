"""Train He (|psi|^2 sampler), then estimate the variational energy on a large independent |psi|^2 sample: it must sit just above
the exact lowest antisymmetric eigenvalue -1.8161 (scratch/he1d_exact.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import vqmc
steps, batch, lr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
t = vqmc.ModelTrainer(system_name='He', learning_rate=lr, box_length=10, num_epochs=steps, batch_size=batch, log_every=10**9)
t.save_dir = '/tmp/wf_var'; t.exact_sampler = True
t0 = time.time(); params, loss = t.start_training(verbose=False); print(f"trained {steps} steps in {time.time()-t0:.1f} s; last-1000 mean loss {np.mean(loss[-1000:]):.4f}")
m = t.psi.model
m.ensure_params(params)
es = []
for seed in range(8):
    x = m.sample(1000 + seed, 1 << 15, exact=True)
    h, ps = m.hamiltonian(x, t.h_fn.protons, return_psi=True)
    es.append((h / (ps + 1e-8)).double().cpu().numpy())
e = np.concatenate(es)
print(f"variational energy on {e.size} |psi|^2 samples: {e.mean():.5f} +- {e.std()/np.sqrt(e.size):.5f}   (exact -1.8161); median {np.median(e):.4f}, std {e.std():.3f}")
