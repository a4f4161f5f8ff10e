"""rocprofv3 --kernel-trace target: 60 captured training steps of 2^17 walkers (He)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveflow_amd import vqmc
n = int(os.environ.get("STEPS", 60))
t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=n, batch_size=1 << 17, log_every=10**9)
t.save_dir = '/tmp/wf_trace_train'; t.exact_sampler = True
t0 = time.time(); t.start_training(verbose=False); print('ms/step (incl. setup)', (time.time() - t0) / n * 1e3)
