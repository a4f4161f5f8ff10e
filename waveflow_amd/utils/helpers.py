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
    if isinstance(t, tuple):
        return tuple(_to_numpy_tree(q) for q in t)
    if isinstance(t, list):
        return [_to_numpy_tree(q) for q in t]
    return _np(t)


def _signed_psi(psi, params, coordinates):
    """psi(sorted coordinates) * (-1)^inversions (helpers.py:55-58)"""
    coordinates = np.asarray(coordinates, dtype=np.float32)
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
