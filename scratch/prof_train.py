import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveflow_amd import vqmc
t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=3000, batch_size=128, log_every=10**9)
t.save_dir = '/tmp/wf_prof_train'; t.exact_sampler = True
t0 = time.time(); t.start_training(verbose=False); print('ms/step', (time.time() - t0) / 3000 * 1e3)
