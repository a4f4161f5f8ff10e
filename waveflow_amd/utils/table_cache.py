"""The reference's on-disk basis-table cache (isplines_jax.py:104-131, msplines_jax.py:84-108, bsplines_jax.py:68-116), written and read with
its file names so that a cache directory is interchangeable between the two code bases:

    <root>/degree_{k}_niknots_{N}_nmp_{n_mesh}_nd_{0..3}.npy                      I- and M-splines, N = number of bases, [N][n_mesh]
    <root>/b_degree_{k}_niknots_{N+1}_nmp_{n_mesh}_nd_{0..3}.npy                  plain B-splines, N bases ("n_knots - k" in the name)
    <root>/ob_degree_{k}_niknots_{N+1}_nmp_{n_mesh}_nd_{0..3}.npy                 orthogonalised B-splines
    <root>/degree_{k}_niknots_{N+1}_nmp_{n_mesh}_{b_to_ob,ob_to_b}.npy            change-of-basis matrices [N][N]

The reference keeps one root per spline family (`./splines/cached_bases/{I,M,B}/`).  Tables come from wf_tables_build (host only, fp64: the
values of the reference's own fixture files, tests/test_oracle.py); nothing here needs a GPU.  The device models build their tables in
memory; this module is for interop with a reference installation (pre-seeding its cache saves its ~40 s per (degree, knots) precompute)."""
import os

import numpy as np

from .. import _lib
from ..core import build_tables

_KINDS = {"I": _lib.SPLINE_I, "M": _lib.SPLINE_M, "B": _lib.SPLINE_B}


def cache_file_names(kind, degree, n_internal_knots, n_mesh=2000):
    """-> dict of the file names (no directory) the reference looks for: {"nd": [4 names]} for I / M; for B also "ob", "b_to_ob", "ob_to_b"."""
    if kind not in _KINDS:
        raise ValueError("kind must be 'I', 'M' or 'B'")
    nb = _lib.check(_lib.lib().wf_tables_build(_KINDS[kind], degree, n_internal_knots, n_mesh, None, None, None), "wf_tables_build")
    if kind == "B":
        stem = f"degree_{degree}_niknots_{nb + 1}_nmp_{n_mesh}"
        return {"nd": [f"b_{stem}_nd_{nd}.npy" for nd in range(4)], "ob": [f"ob_{stem}_nd_{nd}.npy" for nd in range(4)],
                "b_to_ob": f"{stem}_b_to_ob.npy", "ob_to_b": f"{stem}_ob_to_b.npy"}
    return {"nd": [f"degree_{degree}_niknots_{nb}_nmp_{n_mesh}_nd_{nd}.npy" for nd in range(4)]}


def write_cached_bases(root, kind, degree, n_internal_knots, n_mesh=2000):
    """Fill `root` with the tables of one (family, degree, knots, mesh) under the reference's names.  -> list of paths written."""
    os.makedirs(root, exist_ok=True)
    names = cache_file_names(kind, degree, n_internal_knots, n_mesh)
    written = []

    def save(name, arr):
        path = os.path.join(root, name)
        np.save(path, np.ascontiguousarray(arr, dtype=np.float64))
        written.append(path)

    t = build_tables(_KINDS[kind], degree, n_internal_knots, n_mesh)
    for nd in range(4):
        save(names["nd"][nd], t[nd])
    if kind == "B":
        ob, b2o, o2b = build_tables(_lib.SPLINE_OB, degree, n_internal_knots, n_mesh)
        for nd in range(4):
            save(names["ob"][nd], ob[nd])
        save(names["b_to_ob"], b2o)
        save(names["ob_to_b"], o2b)
    return written


def load_cached_bases(root, kind, degree, n_internal_knots, n_mesh=2000):
    """What the reference's init_fun loads from `root`: fp64 [4][N][n_mesh] (for B: plain, orthogonal, b_to_ob, ob_to_b).  Raises
    FileNotFoundError when the cache does not hold this combination."""
    names = cache_file_names(kind, degree, n_internal_knots, n_mesh)
    stack = lambda key: np.stack([np.load(os.path.join(root, n)) for n in names[key]])
    if kind == "B":
        return stack("nd"), stack("ob"), np.load(os.path.join(root, names["b_to_ob"])), np.load(os.path.join(root, names["ob_to_b"]))
    return stack("nd")
