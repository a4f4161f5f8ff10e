"""Soft pin of the maximum-likelihood trainer: the reference's three shipped density-estimation runs (data_submission_apl_ml/
double_circles/*/losses.txt: Adam 1e-4 on 20000 'circles' points) reached, from their first to their last recorded epoch,
    MFlow reg 0.05, 15 knots, prior degree 5: -0.0523 -> -1.1651 (12000 epochs)
    MFlow reg 0.02, 23 knots, prior degree 5: -0.0472 -> -1.1965 (30000 epochs)
    Flow (3 MADE layers, Normal(-0.5)):         1.9807 -> -0.7573 (30000 epochs)
Same settings here, on a fresh draw of the same data distribution."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveflow_amd import benchmark_tests
X = benchmark_tests.get_dataset("circles", 20000, 0.025, 0)
for name, kw, epochs, ref in (("MFlow_0.05_3_5_15", dict(model_type="MFlow", spline_reg=0.05, num_knots=15, prior_spline_degree=5), 12000, (-0.0523, -1.1651)),
                              ("MFlow_0.02_3_5_23", dict(model_type="MFlow", spline_reg=0.02, num_knots=23, prior_spline_degree=5), 30000, (-0.0472, -1.1965)),
                              ("Flow_3", dict(model_type="Flow", spline_reg=0.05, num_knots=15), 30000, (1.9807, -0.7573))):
    t = time.time()
    params, losses = benchmark_tests.train_model(X, epochs, 1000, dataset_name="circles", check_step=10**9, save_dir="/tmp/wf_softpin", ngrid=50,
                                                 num_flow_layer=3, spline_degree=5, prior_num_knots=15, step_size=1e-4, verbose=False, **kw)
    print(f"{name}: {epochs} epochs in {time.time()-t:.1f} s: loss {losses[0]:.4f} -> {losses[-1]:.4f}   (reference {ref[0]:.4f} -> {ref[1]:.4f})")
