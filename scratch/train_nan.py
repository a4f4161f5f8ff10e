import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import vqmc
batch, steps = int(sys.argv[1]), int(sys.argv[2])
t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=steps, batch_size=batch, log_every=10**9)
t.save_dir = f'/tmp/wf_nan_{batch}_{os.environ.get("WF_GRAD_R3", "0")}'
t.exact_sampler = True
t.use_graph = os.environ.get('NOGRAPH') is None
t0 = time.time(); params, loss = t.start_training(verbose=False); dt = time.time() - t0
l = np.asarray(loss[1:], dtype=np.float64)
bad = np.flatnonzero(~np.isfinite(l))
print(f'R3={os.environ.get("WF_GRAD_R3", "0")} batch {batch}: {steps} steps in {dt:.1f} s; first non-finite loss at step {bad[0] if bad.size else None}; losses[:5] {l[:5]}; around: {l[max(bad[0]-3,0):bad[0]+2] if bad.size else l[-3:]}')
