import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import benchmark_tests
X = benchmark_tests.get_dataset("circles", 20000, 0.025, 0)
for seed in range(4):
    params, losses = benchmark_tests.train_model(X, 30000, 1000, model_type="Flow", dataset_name="circles", check_step=10**9, save_dir="/tmp/wf_fs", ngrid=50,
                                                 num_flow_layer=3, step_size=1e-4, verbose=False, seed=seed)
    l = np.asarray(losses)
    print(f"seed {seed}: loss {l[0]:.4f} -> 5k {l[5000]:.4f}, 10k {l[10000]:.4f}, 20k {l[20000]:.4f}, 30k {l[-1]:.4f}")
