"""2-D density-estimation benchmark on one MI355X -- the drop-in of the reference's examples/run_benchmark.py
(the reference script saves the data set and exits before training; here the training call below it runs):

    python examples/run_benchmark.py [--model Flow|IFlow|MFlow] [--dataset circles|halfmoon|gaussian_mixtures] [--epochs N] ...
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from waveflow_amd import benchmark_tests  # noqa: E402   (reference: from waveflow import benchmark_tests)

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="Flow")
ap.add_argument("--dataset", default="circles")
ap.add_argument("--epochs", type=int, default=80000)
ap.add_argument("--check-step", type=int, default=5000)
ap.add_argument("--n-samples", type=int, default=20000)
ap.add_argument("--ngrid", type=int, default=300)
ap.add_argument("--step-size", type=float, default=1e-4)
ap.add_argument("--save-dir", default="./results/benchmarks/")
args = ap.parse_args()

n_samples = args.n_samples
margin = 0.025
input_dim = 2
n_model_sample = 20000
spline_reg = 0.01
spline_degree = 5
num_knots = 15
num_layer = 3
prior_degree = 3
prior_num_knots = 15

X = benchmark_tests.get_dataset(args.dataset, n_samples, margin, 0)
ref_dir = f"{args.save_dir}/{args.dataset}/reference/outputs/"
Path(ref_dir).mkdir(parents=True, exist_ok=True)
orig_sample_file = f"{ref_dir}/values_n{n_samples}.npy"
if not os.path.isfile(orig_sample_file):
    np.save(orig_sample_file, X)
benchmark_tests.train_model(X, args.epochs, n_model_sample, model_type=args.model, dataset_name=args.dataset,
                            check_step=args.check_step, spline_reg=spline_reg, input_dim=input_dim, ngrid=args.ngrid,
                            num_flow_layer=num_layer, num_knots=num_knots, spline_degree=spline_degree,
                            prior_spline_degree=prior_degree, prior_num_knots=prior_num_knots, save_dir=args.save_dir,
                            step_size=args.step_size)
