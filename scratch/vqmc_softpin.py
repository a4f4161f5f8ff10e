"""Soft pin of the VQMC trainer: the reference ships the loss curve of its He run (data_submission_apl_ml/He_1d_L10box_batch256/
loss.npy: batch 256, 100000 epochs, its own sampler).  Medians over 500 epochs starting at the listed epoch:
    1: 6.14   1000: 2.47   2000: 1.55   5000: -0.13   10000: -0.90   20000: -1.38   30000: -1.65   50000: -1.807   70000: -1.814
    90000: -1.91 (below the variational bound -1.8161: the sampler's bias, made.py:88)   99000: -10.4 (diverged)
Same settings here, both samplers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import vqmc
for exact in (False, True):
    t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=100000, batch_size=256, log_every=10**9)
    t.save_dir = '/tmp/wf_softpin_%d' % exact; t.exact_sampler = exact
    t0 = time.time(); params, loss = t.start_training(verbose=False); dt = time.time() - t0
    l = np.asarray(loss[1:], dtype=float)
    print(("|psi|^2 sampler" if exact else "reference sampler") + f": {dt:.1f} s; medians over 500 epochs from epoch")
    print("   " + "  ".join(f"{a}: {np.median(l[a:a+500]):.3f}" for a in (1, 1000, 2000, 5000, 10000, 20000, 30000, 50000, 70000, 90000, 99000)))
