import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import vqmc
def run(steps, batch, lr, exact, tag):
    t = vqmc.ModelTrainer(system_name='He', learning_rate=lr, box_length=10, num_epochs=steps, batch_size=batch, log_every=10**9)
    t.save_dir = f'/tmp/wf_long_{tag}'
    t.exact_sampler = exact
    t0 = time.time()
    params, loss = t.start_training(verbose=False)
    dt = time.time() - t0
    l = np.asarray(loss[1:], dtype=np.float64)
    k = max(steps // 8, 1)
    print(f'{tag}: {steps} steps batch {batch} lr {lr} exact={exact}: {dt:.1f} s ({dt/steps*1e3:.3f} ms/step)')
    print('   window means:', ' '.join(f'{l[a:a+k].mean():.3f}' for a in range(0, steps, k)))
    print('   last 2000: mean %.4f  median %.4f  std %.3f' % (l[-2000:].mean(), np.median(l[-2000:]), l[-2000:].std()))
run(200000, 128, 1e-4, True, 'ref_cfg_exact')
run(60000, 1024, 1e-3, True, 'big_exact')
run(200000, 128, 1e-4, False, 'ref_cfg_refsampler')
