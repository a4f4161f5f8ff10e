"""Reference on-disk parameter formats (helpers.py:39-40: pickle.dump((params, epoch)))."""
import pickle

import numpy as np


class _Unpickler(pickle.Unpickler):
    """Whitelisting unpickler: a reference checkpoint's only non-NumPy global is
    jax._src.array._reconstruct_array(fun, args, arr_state, aval_state); rebuild the wrapped ndarray."""

    def find_class(self, module, name):
        if module == "jax._src.array" and name == "_reconstruct_array":
            def rec(fun, args, arr_state, aval_state):
                a = fun(*args)
                a.__setstate__(arr_state)
                return a
            return rec
        if module.split(".")[0] == "numpy":
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refused global {module}.{name}")


def load_reference_checkpoint(path):
    """-> (params pytree of numpy arrays, epoch)"""
    with open(path, "rb") as f:
        return _Unpickler(f).load()


def unflatten_like(template, flat):
    """Rebuild a pytree shaped like `template` from a flat vector (leaf order)."""
    flat = np.asarray(flat, dtype=np.float32).reshape(-1)
    pos = [0]

    def rec(t):
        if isinstance(t, tuple):
            return tuple(rec(q) for q in t)
        if isinstance(t, list):
            return [rec(q) for q in t]
        a = np.asarray(t)
        out = flat[pos[0]:pos[0] + a.size].reshape(a.shape).copy()
        pos[0] += a.size
        return out

    tree = rec(template)
    if pos[0] != flat.size:
        raise ValueError(f"flat vector has {flat.size} values, template needs {pos[0]}")
    return tree
