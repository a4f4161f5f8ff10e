import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import vqmc
for batch in (128, 256, 4096):
    steps = 20000
    t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=steps, batch_size=batch, log_every=10**9)
    t.save_dir = f'/tmp/wf_quick_{batch}'
    t.exact_sampler = True
    t0 = time.time(); params, loss = t.start_training(verbose=False); dt = time.time() - t0
    l = np.asarray(loss[1:], dtype=np.float64)
    print(f'batch {batch}: {steps} steps in {dt:.1f} s ({dt/steps*1e3:.3f} ms/step incl. first checkpoint); last 2000 median {np.median(l[-2000:]):.4f}')
