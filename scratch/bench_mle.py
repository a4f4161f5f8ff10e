import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import benchmark_tests
X = benchmark_tests.get_dataset("circles", 20000, 0.025, 0)
for mt in ("Flow", "IFlow", "MFlow"):
    t = time.time()
    n = 3000
    params, losses = benchmark_tests.train_model(X, n, 1000, model_type=mt, dataset_name="circles", check_step=10**9, spline_reg=0.01,
                                                 save_dir="/tmp/wf_bench", ngrid=50, num_flow_layer=3, spline_degree=5, num_knots=15,
                                                 step_size=1e-3, verbose=False)
    dt = time.time() - t
    print(f"{mt}: {n} epochs x 20000 points in {dt:.1f} s ({dt/n*1e3:.1f} ms/epoch incl. first checkpoint); loss {losses[0]:.4f} -> {losses[-1]:.4f}")
