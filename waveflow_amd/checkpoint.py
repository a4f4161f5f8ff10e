"""Reference on-disk parameter formats (helpers.py:39-40: pickle.dump((params, epoch)))."""
import pickle

import numpy as np


# The only globals a reference checkpoint needs: the ndarray / dtype reconstructors of NumPy (old and new module paths) and
# jax's array shim.  Anything else -- including the rest of the numpy package, which contains callables that execute code
# (numpy.testing._private.utils.runstring, ...) -- is refused.
_ALLOWED = {
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
}


def _is_reconstructor(fun):
    import numpy as np
    try:
        from numpy._core import multiarray as ma
    except ImportError:   # numpy < 2
        from numpy.core import multiarray as ma
    return fun is ma._reconstruct or fun is np.ndarray


class _Unpickler(pickle.Unpickler):
    """Whitelisting unpickler: exact (module, name) pairs only.  A reference checkpoint's only non-NumPy global is
    jax._src.array._reconstruct_array(fun, args, arr_state, aval_state); it is mapped to "rebuild the wrapped ndarray", and its
    `fun` must itself be NumPy's array reconstructor."""

    def find_class(self, module, name):
        if module == "jax._src.array" and name == "_reconstruct_array":
            def rec(fun, args, arr_state, aval_state):
                if not _is_reconstructor(fun):
                    raise pickle.UnpicklingError("refused reconstructor in jax array record")
                a = fun(*args)
                a.__setstate__(arr_state)
                return a
            return rec
        if (module, name) in _ALLOWED:
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
