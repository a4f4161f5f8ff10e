import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import vqmc
t = vqmc.ModelTrainer(system_name='He', learning_rate=3e-4, box_length=10, num_epochs=40000, batch_size=1024, log_every=10**9)
t.save_dir = '/tmp/wf_ab'; t.exact_sampler = True
params, loss = t.start_training(verbose=False)
m = t.psi.model; m.ensure_params(params)
def stats(nper, reps, tag):
    es, xs = [], []
    for seed in range(reps):
        x = m.sample(9000 + seed, nper, exact=True)
        h, ps = m.hamiltonian(x, t.h_fn.protons, return_psi=True)
        es.append((h / (ps + 1e-8)).double().cpu().numpy()); xs.append(x.cpu().numpy())
    e = np.concatenate(es); x = np.concatenate(xs)
    print(f"{tag}: n={e.size} mean {e.mean():.4f} +- {e.std()/np.sqrt(e.size):.4f} median {np.median(e):.4f} q10 {np.quantile(e,.1):.3f} q90 {np.quantile(e,.9):.3f};  x0 mean {x[:,0].mean():.4f} std {x[:,0].std():.4f}  x1 mean {x[:,1].mean():.4f} std {x[:,1].std():.4f}")
    return x
xa = stats(32768, 10, "wave sampler      ")
xb = stats(40000, 8, "one-lane sampler  ")
from scipy import stats as st
for c in range(2):
    print("KS col", c, st.ks_2samp(xa[:, c], xb[:, c]))
def outl(nper, reps, tag):
    tot = 0; big = []
    for seed in range(reps):
        x, lat = m.sample(9000 + seed, nper, return_latent=True, exact=True)
        h, ps = m.hamiltonian(x, t.h_fn.protons, return_psi=True)
        e = (h / (ps + 1e-8)).double().cpu().numpy()
        idx = np.where(np.abs(e) > 50)[0]
        tot += e.size
        for i in idx[:50]:
            big.append((float(e[i]), lat[i].cpu().numpy().round(5).tolist(), x[i].cpu().numpy().round(4).tolist(), float(ps[i])))
    print(tag, "outliers |E|>50:", len(big), "of", tot)
    for b in sorted(big, key=lambda q: -abs(q[0]))[:8]: print("   ", b)
outl(32768, 10, "wave")
outl(40000, 8, "one-lane")
