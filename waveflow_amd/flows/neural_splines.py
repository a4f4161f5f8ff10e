"""Rational-quadratic spline bijector on the HIP path (reference: flows/bijections/neural_splines.py:16-184).

The reference module is dead code (it calls the removed jax.ops API) and has no fixtures: parity is against the
formulas as restated in oracle/wf_oracle.c.  Same argument names as the reference; tensors are torch.cuda or numpy.
"""
import ctypes

import numpy as np

from .. import _lib

DEFAULT_MIN_BIN_WIDTH = DEFAULT_MIN_BIN_HEIGHT = DEFAULT_MIN_DERIVATIVE = 1e-3


def _call(inputs, uw, uh, ud, inverse, left, right, bottom, top, return_bin_idx):
    import torch
    was_numpy = not hasattr(inputs, "detach")
    dev = "cuda"
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32) if was_numpy else a).to(device=dev, dtype=torch.float32).contiguous()
    x, uw, uh, ud = t(inputs), t(uw), t(uh), t(ud)
    shape = x.shape
    K = uw.shape[-1]
    N = x.numel()
    if uw.numel() != N * K or uh.numel() != N * K or ud.numel() not in (N * (K - 1), N * (K + 1)):
        raise ValueError("shape mismatch between inputs and spline parameters")
    y, ld = torch.empty_like(x), torch.empty_like(x)
    b = torch.empty(shape, device=dev, dtype=torch.int32) if return_bin_idx else None
    P = lambda a: ctypes.c_void_p(a.data_ptr()) if a is not None and a.numel() else None
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().wf_rqs_fwd(P(x), P(uw), P(uh), P(ud), N, K, ud.numel() // max(N, 1) if N else K - 1, int(bool(inverse)),
                                     float(left), float(right), float(bottom), float(top), P(y), P(ld), P(b), stream), "wf_rqs_fwd")
    out = (y, ld, b) if return_bin_idx else (y, ld)
    return tuple(o.cpu().numpy() for o in out) if was_numpy else out


def RQS(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=False, left=0.0, right=1.0,
        bottom=0.0, top=1.0, min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
        min_derivative=DEFAULT_MIN_DERIVATIVE, return_bin_idx=False):
    """neural_splines.py:74-184; unnormalized_derivatives has K+1 columns."""
    if (min_bin_width, min_bin_height, min_derivative) != (1e-3, 1e-3, 1e-3):
        raise NotImplementedError("the HIP kernel is built with the reference's default minima (1e-3)")
    if unnormalized_derivatives.shape[-1] != unnormalized_widths.shape[-1] + 1:
        raise ValueError("RQS needs K+1 derivatives")
    return _call(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse, left, right, bottom, top,
                 return_bin_idx)


def unconstrained_RQS(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=False, tail_bound=1.0,
                      min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                      min_derivative=DEFAULT_MIN_DERIVATIVE, return_bin_idx=False):
    """neural_splines.py:16-71; unnormalized_derivatives has K-1 columns; identity outside +-tail_bound."""
    if (min_bin_width, min_bin_height, min_derivative) != (1e-3, 1e-3, 1e-3):
        raise NotImplementedError("the HIP kernel is built with the reference's default minima (1e-3)")
    if unnormalized_derivatives.shape[-1] != unnormalized_widths.shape[-1] - 1:
        raise ValueError("unconstrained_RQS needs K-1 derivatives")
    return _call(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse, -tail_bound, tail_bound,
                 -tail_bound, tail_bound, return_bin_idx)
