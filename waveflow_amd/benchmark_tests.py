"""waveflow.benchmark_tests call surface on the HIP path (reference: benchmark_tests.py:14-148): 2-D density estimation.

Maximum-likelihood training of Flow / IFlow / MFlow on a toy data set: every epoch evaluates -mean log_pdf over the whole
target set and its parameter gradient (wf_logpdf_fwd + wf_logpdf_vjp with w = -1/N), then one Adam step (step size 1e-4).
"""
import json
import os
from datetime import datetime
from pathlib import Path

import numpy as np

from . import _lib, flows
from .model_factory import get_masked_transform
from .utils import helpers
from .vqmc import adam


def get_dataset(dataset_name, n_samples, margin, rng=None):
    """benchmark_tests.py:14-46.  X in [margin, 1 - margin]^2 (MinMaxScaler)."""
    from sklearn import datasets, mixture, preprocessing
    g = flows.as_generator(0 if rng is None else rng)
    if dataset_name == 'gaussian_mixtures':
        target = datasets.make_blobs(center_box=(-1, 1), cluster_std=0.1, random_state=3)[0]
        gmm = mixture.GaussianMixture(3, random_state=0)
        gmm.fit(target)
        comp = g.choice(3, size=10000, p=gmm.weights_ / gmm.weights_.sum())
        X = np.stack([g.multivariate_normal(gmm.means_[c], gmm.covariances_[c]) for c in comp])
    elif dataset_name == 'halfmoon':
        X, _ = datasets.make_moons(n_samples=n_samples, noise=.05, random_state=int(g.integers(1 << 31)))
    elif dataset_name == 'circles':
        X, _ = datasets.make_circles(n_samples=n_samples, noise=.05, factor=0.5, random_state=int(g.integers(1 << 31)))
    else:
        raise ValueError(f"unknown dataset {dataset_name!r}")
    scaler = preprocessing.MinMaxScaler(feature_range=(margin, 1 - margin))
    return scaler.fit_transform(X)


def get_model(model_type, spline_reg, spline_degree=3, num_knots=15, num_layers=5, reverse_tol=1e-6, prior_spline_degree=3,
              prior_num_knots=15):
    """benchmark_tests.py:50-79"""
    def imade_stack():
        layers = []
        for _ in range(num_layers):
            layers += [flows.IMADE(get_masked_transform(), spline_degree=spline_degree, n_internal_knots=num_knots,
                                   spline_regularization=spline_reg, reverse_fun_tol=reverse_tol), flows.Reverse()]
        return flows.Serial(*layers)

    if model_type == 'Flow':
        layers = []
        for _ in range(num_layers):
            layers += [flows.MADE(get_masked_transform(return_simple_masked_transform=True)), flows.Reverse()]
        return flows.Flow(flows.Serial(*layers), flows.Normal(-0.5))
    if model_type == 'IFlow':
        return flows.Flow(imade_stack(), flows.Uniform(), prior_support=(0.0, 1.0))
    if model_type == 'MFlow':
        return flows.MFlow(imade_stack(), get_masked_transform(), spline_degree=prior_spline_degree, n_internal_knots=prior_num_knots)
    raise ValueError('No supported model type selected.')


def loss(params, target, log_pdf):
    """benchmark_tests.py:84-87"""
    return float(-helpers._np(log_pdf(params, target)).astype(np.float64).mean())


def loss_and_grad(params, target, log_pdf):
    """value and gradient of `loss` (grad(loss) at benchmark_tests.py:100): one forward and one reverse launch."""
    model = log_pdf.model
    model.ensure_params(params)
    t, _ = model._to_dev(target)
    lp, grad = model.logpdf_loss_grad(t, -1.0 / t.shape[0])     # log_pdf of every point and the gradient of their mean, one sweep pair
    s = model.block_sums(lp).cpu().tolist()
    return -s[0] / s[2], grad


def _run_directory(save_dir, dataset_name, model_type, spline_reg, num_flow_layer, spline_degree, num_knots):
    """Directory naming of benchmark_tests.py:112-119 (a second run into the same place gets a time-stamped sub-directory)."""
    tag = f"{model_type}_{num_flow_layer}" if model_type == "Flow" else f"{model_type}_{spline_reg}_{num_flow_layer}_{spline_degree}_{num_knots}/"
    run_dir = f"{save_dir}/{dataset_name}/{tag}"
    if os.path.exists(f"{run_dir}/outputs/"):
        run_dir = f"{run_dir}/{datetime.now().strftime('%M-%D-%H')}"
    Path(run_dir).mkdir(parents=True, exist_ok=True)
    return run_dir


def train_model(target, num_epochs, n_model_sample, model_type='IFlow', dataset_name='halfmoon', check_step=5000, spline_reg=0.1,
                input_dim=2, save_dir="./results/benchmarks/", ngrid=300, num_flow_layer=3, spline_degree=5, num_knots=23,
                prior_spline_degree=3, prior_num_knots=15, step_size=1e-4, seed=0, verbose=True):
    """benchmark_tests.py:90-148: same artefacts (system_info.json, outputs/*.npy, losses.txt, ...).  -> (params, losses)"""
    import torch
    g = np.random.default_rng(seed)
    params, log_pdf, sample = get_model(model_type, spline_reg, spline_degree=spline_degree, num_layers=num_flow_layer, num_knots=num_knots,
                                        prior_spline_degree=prior_spline_degree, prior_num_knots=prior_num_knots)(int(g.integers(1 << 31)), input_dim)
    opt_init, opt_update, get_params = adam(step_size=step_size, model=log_pdf.model)   # parameters and Adam state stay on the GPU
    state = opt_init(params)
    x_dev = torch.as_tensor(np.asarray(target, dtype=np.float32)).cuda(log_pdf.model.device)   # the target set stays in HBM

    run_dir = _run_directory(save_dir, dataset_name, model_type, spline_reg, num_flow_layer, spline_degree, num_knots)
    with open(f"{run_dir}/system_info.json", "w") as f:
        json.dump(dict(model_type=model_type, dataset_name=dataset_name, splines_regulation=spline_reg, flow_spline_degree=spline_degree,
                       flow_spline_num_knots=num_knots, prior_spline_degree=prior_spline_degree), f)

    losses = [loss(params, x_dev, log_pdf)]
    metrics = ([], [], [])   # KDE KL divergences, KDE Hellinger distances, reconstruction distances
    model = log_pdf.model
    # every epoch is the same sequence of launches on resident data: capture it once (wf_mle_train_step) and replay it; the
    # host reads the loss ring at checkpoints and every `ring` epochs
    ring = 256
    st = model.make_train_state(state.x, state.m, state.v, 1, ring_len=ring, defer_eval_tables=True)   # refreshed before every evaluation below
    model.set_params_device(state.x)
    nbytes = _lib.check(_lib.lib().wf_mle_train_step_workspace_bytes(model._h, int(x_dev.shape[0])), "wf_mle_train_step_workspace_bytes")
    st["ws"] = model._workspace(nbytes, x_dev.device)
    side = torch.cuda.Stream(device=model.device)
    side.wait_stream(torch.cuda.current_stream(model.device))
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            model.mle_train_step(st, x_dev, step_size)
    torch.cuda.current_stream(model.device).wait_stream(side)
    fetched = [0]

    def fetch(upto):
        torch.cuda.synchronize(model.device)
        r = st["ring"].cpu().numpy()
        losses.extend(float(-r[e % ring, 0] / r[e % ring, 2]) for e in range(fetched[0] + 1, upto + 1))
        fetched[0] = upto

    for epoch in range(1, num_epochs + 1):
        if epoch == 1 or epoch % check_step == 0:
            fetch(epoch - 1)
            state.version += 1
            helpers.make_checkpoint_benchmark(int(g.integers(1 << 31)), get_params(state), log_pdf, sample, losses, *metrics,
                                              n_model_sample=n_model_sample, save_dir=run_dir, epoch=epoch, ngrid=ngrid)
            model.set_params_device(state.x)
        # (the reference permutes the target rows here; the full-batch mean does not depend on their order)
        graph.replay()
        if epoch % ring == 0:
            fetch(epoch)
        if verbose and epoch % check_step == 0:
            fetch(epoch)
            print(f"Epoch {epoch} | loss: {losses[-1]}")
    fetch(num_epochs)
    state.version += 1
    model.set_params_device(state.x)   # every image, evaluation tables included, holds the final parameters
    return get_params(state), losses
