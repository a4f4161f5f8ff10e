#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference).  It never copies
reference source: it (a) repacks data files the reference ships, and
(b) imports the reference's NumPy-only modules (waveflow.splines.splines_np,
waveflow.splines.ortho_splines -- the JAX modules are not importable here:
ModuleNotFoundError: jax) and records input/output vectors.

Outputs (all small, data only):
  ref_tables_k5_n16.npz   the reference's own test fixtures
                          (waveflow/tests/splines/cached_bases/{I,B}/*.npy):
                          I nd0..3 and plain-B nd0..3 complete (fp64), the two
                          20x20 change-of-basis matrices, and the orthogonal-B
                          tables on every 16th mesh column.
  ref_probes.npz          splines_np.{M,I,B} evaluated by the reference itself
                          on sparse mesh columns for the (k, knots) pairs the
                          shipped configs use; ortho_splines.gram_schmidt_symm
                          output for the (k=6, 23 knot) B table (sub-sampled).
  he_checkpoint.npz       the He checkpoint (pickle -> flat fp32 leaves).
  he_golden.npz           psi grids / cuts / samples the reference wrote for
                          that checkpoint (helpers.py:52-89).
  circles_x256.npy        first 256 rows of the double-circles dataset.
"""
import io
import os
import pickle
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)


def pack_reference_fixture_tables():
    root = f"{REF}/waveflow/tests/splines/cached_bases"
    d = {}
    for nd in range(4):
        d[f"I_nd{nd}"] = np.load(f"{root}/I/degree_5_niknots_21_nmp_2000_nd_{nd}.npy")
        d[f"B_nd{nd}"] = np.load(f"{root}/B/b_degree_5_niknots_21_nmp_2000_nd_{nd}.npy")
        ob = np.load(f"{root}/B/ob_degree_5_niknots_21_nmp_2000_nd_{nd}.npy")
        cols = np.unique(np.concatenate([np.arange(0, 2000, 16), [1, 1998, 1999]]))
        d["OB_cols"] = cols
        d[f"OB_nd{nd}_sub"] = ob[:, cols]
    d["b_to_ob"] = np.load(f"{root}/B/degree_5_niknots_21_nmp_2000_b_to_ob.npy")
    d["ob_to_b"] = np.load(f"{root}/B/degree_5_niknots_21_nmp_2000_ob_to_b.npy")
    np.savez_compressed(f"{OUT}/ref_tables_k5_n16.npz", **d)


def knots_I(k, n):  # isplines_jax.py:91-93
    t = np.linspace(0, 1, n)
    t = np.repeat(t, ((t == t[0]) * (k + 1)).clip(min=1))
    return np.repeat(t, ((t == t[-1]) * (k + 1)).clip(min=1))


def knots_B(k, n):  # bsplines_jax.py:58-60
    t = np.linspace(0, 1, n)
    t = np.repeat(t, ((t == t[0]) * k + 1).clip(min=1))
    return np.repeat(t, ((t == t[-1]) * k + 1).clip(min=1))


def knots_M(k, n):  # msplines_jax.py:72-74
    t = np.linspace(0, 1, n)
    t = np.repeat(t, ((t == t[0]) * k).clip(min=1))
    return np.repeat(t, ((t == t[-1]) * k).clip(min=1))


def probes():
    from waveflow.splines import splines_np as ref_np
    from waveflow.splines import ortho_splines as ref_ortho

    n_mesh = 2000
    mesh = np.linspace(0, 1, n_mesh)
    rng = np.random.default_rng(0)
    cols = np.unique(np.concatenate([[0, 1, 2, 90, 91, 92, 999, 1000, 1817, 1818, 1997, 1998, 1999],
                                     rng.integers(0, n_mesh, 28)]))
    d = {"cols": cols}
    # (kind, k, n_internal_knots): He run (6,23); benchmark MFlow runs (5,15),(5,23) I and (3,15),(5,15) M
    cases = [("I", 6, 23), ("B", 6, 23), ("M", 6, 23), ("I", 5, 15), ("I", 5, 23), ("M", 3, 15), ("M", 5, 15),
             ("I", 3, 9), ("M", 3, 9), ("B", 3, 9), ("I", 5, 9), ("M", 5, 9)]
    for kind, k, n in cases:
        if kind == "I":
            t = knots_I(k, n); nb = len(t) - k
            f = lambda x, i, nd: ref_np.I(x, k, i, t, k + 1, n_derivatives=nd)
        elif kind == "B":
            t = knots_B(k, n); nb = len(t) - k - 1
            f = lambda x, i, nd: ref_np.B(x, k, i, t, k, n_derivatives=nd)
        else:
            t = knots_M(k, n); nb = len(t) - k
            f = lambda x, i, nd: ref_np.M(x, k, i, t, k, n_derivatives=nd)
        vals = np.zeros((4, nb, len(cols)))
        for nd in range(4):
            for i in range(nb):
                for c, m in enumerate(cols):
                    vals[nd, i, c] = f(mesh[m], i, nd)
        d[f"{kind}_k{k}_n{n}"] = vals

    # reference Gram-Schmidt on the full (k=6, 23 knots) plain-B nd0 table
    k, n = 6, 23
    t = knots_B(k, n); nb = len(t) - k - 1
    Bfull = np.array([[ref_np.B(x, k, i, t, k, n_derivatives=0) for x in mesh] for i in range(nb)])
    ob = ref_ortho.gram_schmidt_symm(Bfull.T).T
    sub = np.unique(np.concatenate([np.arange(0, 2000, 16), [1, 1998, 1999]]))
    d["gs_B_k6_n23_cols"] = sub
    d["gs_B_k6_n23_in_sub"] = Bfull[:, sub]
    d["gs_B_k6_n23_out_sub"] = ob[:, sub]
    d["gs_B_k6_n23_out_rowsum"] = ob.sum(-1)
    d["gs_B_k6_n23_out_gram"] = ob @ ob.T
    # a tiny dense case for the Gram-Schmidt restatement
    a = np.random.default_rng(1).normal(size=(12, 6)); a = np.abs(a)
    d["gs_small_in"] = a
    # symm_ortho2v asserts 0<=ovlp<=1, so feed a non-negative, well-conditioned matrix
    d["gs_small_out"] = ref_ortho.gram_schmidt_symm(a)
    d["gs_l2r_small_out"] = ref_ortho.gram_schmidt_l2r(a)
    np.savez_compressed(f"{OUT}/ref_probes.npz", **d)


class _Unpickler(pickle.Unpickler):
    """Whitelisting unpickler: the checkpoint's only non-NumPy global is
    jax._src.array._reconstruct_array(fun, args, arr_state, aval_state)."""

    def find_class(self, module, name):
        if module == "jax._src.array" and name == "_reconstruct_array":
            def rec(fun, args, arr_state, aval_state):
                a = fun(*args)
                a.__setstate__(arr_state)
                return a
            return rec
        if module.split(".")[0] == "numpy":
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refused {module}.{name}")


def leaves(p, out):
    if isinstance(p, (tuple, list)):
        for q in p:
            leaves(q, out)
    else:
        out.append(np.asarray(p))
    return out


def he():
    root = f"{REF}/data_submission_apl_ml/He_1d_L10box_batch256"
    with open(f"{root}/checkpoints", "rb") as f:
        params, epoch = _Unpickler(f).load()
    ls = leaves(params, [])
    assert all(a.dtype == np.float32 for a in ls)
    np.savez_compressed(f"{OUT}/he_checkpoint.npz", epoch=epoch,
                        flat=np.concatenate([a.reshape(-1) for a in ls]),
                        shapes=np.array([list(a.shape) + [0] * (2 - a.ndim) for a in ls]),
                        ndims=np.array([a.ndim for a in ls]))
    e = 100000
    np.savez_compressed(
        f"{OUT}/he_golden.npz",
        psi_grid=np.load(f"{root}/outputs/wavefunctions_2d/values_epoch{e}.npy"),
        onproton_coord=np.load(f"{root}/outputs/density_1e/onproton_coord_epoch{e}.npy"),
        onproton_values=np.load(f"{root}/outputs/density_1e/onproton_values_epoch{e}.npy"),
        random_coord=np.load(f"{root}/outputs/density_1e/random_coord_epoch{e}.npy"),
        random_values=np.load(f"{root}/outputs/density_1e/random_values_epoch{e}.npy"),
        sample_points=np.load(f"{root}/outputs/sample_points/values_epoch{e}.npy"),
    )
    x = np.load(f"{REF}/data_submission_apl_ml/double_circles/reference/outputs/values_n20000.npy")
    np.save(f"{OUT}/circles_x256.npy", x[:256])


if __name__ == "__main__":
    pack_reference_fixture_tables()
    probes()
    he()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(f"{OUT}/{f}"))
