import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import benchmark_tests
X = benchmark_tests.get_dataset("circles", 20000, 0.025, 0)
for mt in sys.argv[1:]:
    params, losses = benchmark_tests.train_model(X, 300, 1000, model_type=mt, dataset_name="circles", check_step=10**9, spline_reg=0.01,
                                                 save_dir="/tmp/wf_bench2", ngrid=50, num_flow_layer=3, spline_degree=5, num_knots=15,
                                                 step_size=1e-3, verbose=False)
    print(mt, len(losses), losses[:4], losses[-1])
