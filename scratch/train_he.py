import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import vqmc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
exact = (sys.argv[4] == 'exact') if len(sys.argv) > 4 else True
t = vqmc.ModelTrainer(system_name='He', learning_rate=lr, box_length=10, num_epochs=steps, batch_size=batch, log_every=10**9)
t.save_dir = '/tmp/wf_train_he'
t.exact_sampler = exact
t0 = time.time()
params, loss = t.start_training(verbose=False)
dt = time.time() - t0
l = np.asarray(loss[1:], dtype=np.float64)
print(f'{steps} steps batch {batch} lr {lr} exact={exact}: {dt:.1f} s  ({dt/steps*1e3:.2f} ms/step)')
for a in range(0, steps, max(steps // 10, 1)):
    print(a, np.round(l[a:a + max(steps // 10, 1)].mean(), 4))
