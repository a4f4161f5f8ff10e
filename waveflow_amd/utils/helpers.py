"""waveflow.utils.helpers call surface: the on-disk artefacts of a VQMC run (reference: utils/helpers.py:13-89).

File names, array shapes and dtypes follow the reference so that runs are interchangeable:
    {save_dir}/checkpoints                         pickle((params, epoch))            helpers.py:39-40
    {save_dir}/loss.npy, energies.npy                                                   :42-43
    outputs/wavefunctions_2d/values_epoch{N}.npy   psi on a ngrid x ngrid grid, signed  :50-59
    outputs/density_1e/{random,onproton}_{values,coord}_epoch{N}.npy                    :61-84
    outputs/sample_points/values_epoch{N}.npy      nsample samples                      :86-89
"""
import pickle
from pathlib import Path

import numpy as np

from . import physics
from .coordinates import get_num_inversion_count


def make_result_dirs(save_dir):
    """helpers.py:13-30"""
    for sub in ("figures/eigenfunctions", "figures/densities_random", "figures/densities_on_proton", "outputs/wavefunctions_2d",
                "outputs/sample_points", "outputs/density_1e"):
        Path(f"{save_dir}/{sub}").mkdir(parents=True, exist_ok=True)


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def _to_numpy_tree(t):
    if hasattr(t, "tree") and hasattr(t, "flat"):   # core.DeviceParams
        return _to_numpy_tree(t.tree())
    if isinstance(t, tuple):
        return tuple(_to_numpy_tree(q) for q in t)
    if isinstance(t, list):
        return [_to_numpy_tree(q) for q in t]
    return _np(t)


def _signed_psi(psi, params, coordinates):
    """psi(sorted coordinates) * (-1)^inversions (helpers.py:55-58).  A closure of this package sorts and signs on the device
    (wf_psi_antisym_fwd); any other psi gets the reference's host sequence."""
    coordinates = np.asarray(coordinates, dtype=np.float32)
    model = getattr(psi, "model", None)
    if model is not None and hasattr(model, "psi_antisym"):
        model.ensure_params(params)
        return np.asarray(_np(model.psi_antisym(coordinates)), dtype=np.float32)
    inv = get_num_inversion_count(coordinates)
    z = _np(psi(params, np.sort(coordinates, axis=-1)))
    return (z * ((-1.0) ** inv)).astype(np.float32)


def create_checkpoint_wavefunc(rng, save_dir, psi, sample, params, epoch, loss, energies, system_dict, ngrid=100, nsample=250):
    """helpers.py:33-89.  2 electrons in 1 space dimension, as the reference's grid code assumes."""
    make_result_dirs(save_dir)
    with open(f"{save_dir}/checkpoints", "wb") as f:
        pickle.dump((_to_numpy_tree(params), epoch), f)
    np.save(f"{save_dir}/loss.npy", np.asarray([float(_np(v)) for v in loss]))
    np.save(f"{save_dir}/energies.npy", np.asarray([[float(_np(v)) for v in e] for e in energies]) if len(energies) else np.zeros((0, 1)))
    box_length = system_dict["box_length"]
    n_particle = system_dict["n_particle"]
    n_space_dimension = system_dict["n_space_dimension"]
    protons, _ = physics.system_catalogue[n_space_dimension][system_dict["system_name"]]

    y, x = np.meshgrid(np.linspace(-box_length, box_length, ngrid), np.linspace(-box_length, box_length, ngrid))
    coordinates = np.stack([x, y], axis=-1).reshape(-1, 2)
    np.save(f"{save_dir}/outputs/wavefunctions_2d/values_epoch{epoch}.npy", _signed_psi(psi, params, coordinates))

    one = f"{save_dir}/outputs/density_1e"
    xr = np.repeat(_np(sample(rng, params, 1)).astype(np.float32), ngrid, axis=0)
    xr[:, 0] = np.linspace(-box_length, box_length, ngrid)
    np.save(f"{one}/random_values_epoch{epoch}.npy", _signed_psi(psi, params, xr))
    np.save(f"{one}/random_coord_epoch{epoch}.npy", xr)
    xp = np.repeat(np.ones((1, n_particle * n_space_dimension), np.float32) * protons[0], ngrid, axis=0)
    xp[:, 0] = np.linspace(-box_length, box_length, ngrid)
    np.save(f"{one}/onproton_values_epoch{epoch}.npy", _signed_psi(psi, params, xp))
    np.save(f"{one}/onproton_coord_epoch{epoch}.npy", xp)

    np.save(f"{save_dir}/outputs/sample_points/values_epoch{epoch}.npy", _np(sample(rng, params, nsample)).astype(np.float32))


def make_checkpoint_benchmark(split_rng, params, log_pdf, sample, losses, kde_kl_divergences, kde_hellinger_distances,
                              reconstruction_distances, n_model_sample=5000, save_dir='./results/benchmarks/', epoch=0, ngrid=300):
    """helpers.py:170-214: pdf on the [0,1]^2 grid, model samples, their Gaussian KDE (bandwidth 0.01), the KL / Hellinger
    figures between the two grids and the sample -> latent -> sample reconstruction distance; same file names."""
    from sklearn.neighbors import KernelDensity
    output_dir = f"{save_dir}/outputs/"
    Path(output_dir).mkdir(parents=True, exist_ok=True)
    x = np.linspace(0.0, 1.0, ngrid)
    xv, yv = np.meshgrid(x, x)
    grid = np.stack([xv.reshape(-1), yv.reshape(-1)], axis=-1).astype(np.float32)
    log_pdf_grid = _np(log_pdf(params, grid)).astype(np.float64).reshape(ngrid, ngrid)
    pdf_grid = np.exp(log_pdf_grid)
    np.save(f"{output_dir}/pdf_grid_epoch{epoch}.npy", pdf_grid)

    model_samples, original_samples = sample(split_rng, params, num_samples=n_model_sample, return_original_samples=True)
    model_samples, original_samples = _np(model_samples), _np(original_samples)
    np.save(f"{output_dir}/samples_epoch{epoch}.npy", model_samples)

    kde = KernelDensity(kernel='gaussian', bandwidth=0.01, rtol=0.1).fit(model_samples)
    log_pdf_grid_kde = kde.score_samples(grid).reshape(ngrid, ngrid)
    pdf_grid_kde = np.exp(log_pdf_grid_kde)
    np.save(f"{output_dir}/kde_pdf_grid_epoch{epoch}.npy", pdf_grid_kde)
    kde_kl_divergences.append(float((pdf_grid * (log_pdf_grid - log_pdf_grid_kde)).mean()))
    kde_hellinger_distances.append(float(((np.sqrt(pdf_grid) - np.sqrt(pdf_grid_kde)) ** 2).mean()))

    _, reconstructed = log_pdf(params, model_samples, return_sample=True)
    reconstruction_distances.append(float(np.linalg.norm(original_samples - _np(reconstructed), axis=-1).mean()))

    np.savetxt(f'{save_dir}/losses.txt', np.asarray(losses, dtype=np.float64))
    np.savetxt(f'{save_dir}/kl_divergences.txt', kde_kl_divergences)
    np.savetxt(f'{save_dir}/hellinger_divergences.txt', kde_hellinger_distances)
    np.savetxt(f'{save_dir}/reconstruction_distances.txt', reconstruction_distances)


# ---- small host-side statistics helpers of the reference (helpers.py:121-166), numpy only
def moving_average(running_average, new_data, beta):
    """Exponential moving average step (helpers.py:121-122)."""
    return running_average - beta * (running_average - new_data)


def uniform_sliding_average(data, window):
    """Trailing mean over `window` entries along the last axis, the head padded with its first value (helpers.py:126-134)."""
    data = np.asarray(data, dtype=float)
    pad = [(0, 0)] * (data.ndim - 1) + [(window - 1, 0)]
    c = np.cumsum(np.pad(data, pad, mode='edge'), axis=-1)
    c[..., window:] = c[..., window:] - c[..., :-window]
    return c[..., window - 1:] / window


def uniform_sliding_stdev(data, window):
    """Trailing standard deviation over `window` entries along the last axis, the head reflect-padded (helpers.py:137-146)."""
    data = np.asarray(data, dtype=float)
    pad = [(0, 0)] * (data.ndim - 1) + [(window - 1, 0)]
    p = np.pad(data, pad, mode='reflect')
    v = np.lib.stride_tricks.sliding_window_view(p, window, axis=-1)
    return v.std(axis=-1)


def binary_search(func, low=0.0, high=1.0, tol=1e-3):
    """Bisection with the reference's stopping rule (helpers.py:150-166): returns the lower end of the final bracket.
    (On the device this loop is wf_inverse_fwd; this host version serves small scripts.)"""
    while True:
        mid = 0.5 * (low + high)
        if not ((low + tol / 2 < mid) and (mid < high - tol / 2)):
            return low
        if func(mid) > 0:
            high = mid
        else:
            low = mid
