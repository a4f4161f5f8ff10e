"""2-D density-estimation benchmark on one MI355X -- the counterpart of the reference's examples/run_benchmark.py (which
saves the data set and exits before its training call; here the training runs).  Defaults = the reference's settings:

    python examples/run_benchmark.py [--model Flow|IFlow|MFlow] [--dataset circles|halfmoon|gaussian_mixtures] [--epochs N] ...
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from waveflow_amd import benchmark_tests  # noqa: E402   (reference: from waveflow import benchmark_tests)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Flow", choices=["Flow", "IFlow", "MFlow"])
    ap.add_argument("--dataset", default="circles")
    ap.add_argument("--epochs", type=int, default=80000)
    ap.add_argument("--check-step", type=int, default=5000)
    ap.add_argument("--n-samples", type=int, default=20000, help="size of the training set")
    ap.add_argument("--n-model-sample", type=int, default=20000, help="model samples drawn at every checkpoint")
    ap.add_argument("--margin", type=float, default=0.025)
    ap.add_argument("--ngrid", type=int, default=300)
    ap.add_argument("--spline-reg", type=float, default=0.01)
    ap.add_argument("--spline-degree", type=int, default=5)
    ap.add_argument("--num-knots", type=int, default=15)
    ap.add_argument("--num-layer", type=int, default=3)
    ap.add_argument("--prior-degree", type=int, default=3)
    ap.add_argument("--prior-num-knots", type=int, default=15)
    ap.add_argument("--step-size", type=float, default=1e-4)
    ap.add_argument("--save-dir", default="./results/benchmarks/")
    return ap.parse_args()


def main():
    a = parse()
    data = benchmark_tests.get_dataset(a.dataset, a.n_samples, a.margin, 0)
    # the reference keeps the training set next to the results
    ref_dir = Path(a.save_dir) / a.dataset / "reference" / "outputs"
    ref_dir.mkdir(parents=True, exist_ok=True)
    target_file = ref_dir / f"values_n{a.n_samples}.npy"
    if not target_file.is_file():
        np.save(target_file, data)
    benchmark_tests.train_model(data, a.epochs, a.n_model_sample, model_type=a.model, dataset_name=a.dataset, check_step=a.check_step,
                                spline_reg=a.spline_reg, input_dim=2, ngrid=a.ngrid, num_flow_layer=a.num_layer, num_knots=a.num_knots,
                                spline_degree=a.spline_degree, prior_spline_degree=a.prior_degree, prior_num_knots=a.prior_num_knots,
                                save_dir=a.save_dir, step_size=a.step_size)


if __name__ == "__main__":
    main()
