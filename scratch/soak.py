"""Long runs of the captured training step: no non-finite loss, energy at the variational bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import vqmc
for batch, steps, exact, lr in ((128, 600000, True, 1e-4), (4096, 150000, True, 1e-4), (256, 300000, False, 1e-4)):
    t = vqmc.ModelTrainer(system_name='He', learning_rate=lr, box_length=10, num_epochs=steps, batch_size=batch, log_every=10**9)
    t.save_dir = f'/tmp/wf_soak_{batch}_{int(exact)}'; t.exact_sampler = exact
    t0 = time.time(); params, loss = t.start_training(verbose=False); dt = time.time() - t0
    l = np.asarray(loss[1:], dtype=np.float64)
    k = steps // 6
    print(f"batch {batch} steps {steps} exact={exact}: {dt:.1f} s ({dt/steps*1e3:.3f} ms/step); non-finite losses {int((~np.isfinite(l)).sum())}; "
          f"window medians {' '.join(f'{np.median(l[a:a+k]):.4f}' for a in range(0, steps, k))}; min {np.nanmin(l):.2f} max {np.nanmax(l):.2f}", flush=True)
