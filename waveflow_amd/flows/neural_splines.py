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


def FCNN(out_dim, hidden_dim):
    """neural_splines.py:187-188: the conditioner architecture (Dense, Tanh, Dense, Tanh, Dense) as a description."""
    return ("fcnn", int(out_dim), int(hidden_dim))


class NeuralSplineCoupling:
    """neural_splines.py:244-300.  init_fun(rng, dim) -> (params, direct_fun, inverse_fun); params = (f1_params, f2_params) in
    stax layout [(W, b), (), (W, b), (), (W, b)].  The reference's closures ignore their `params` argument and always use the
    initial networks (:254, :261); here the argument is honoured."""

    def __init__(self, K=5, B=3, hidden_dim=8, network=FCNN):
        if network is not FCNN:
            raise NotImplementedError("NeuralSplineCoupling is built with the reference's FCNN conditioner")
        self.K, self.B, self.hidden = int(K), float(B), int(hidden_dim)
        from . import NSCSpec
        self.spec = NSCSpec(self.K, self.B, self.hidden)   # lets flows.Serial / flows.Flow fuse a stack of these into one model

    def init_params(self, rng, dim):
        from . import as_generator
        if dim % 2:
            raise ValueError("NeuralSplineCoupling needs an even number of dimensions")
        g = as_generator(rng)
        dh, per = dim // 2, 3 * self.K - 1
        return (self._init_net(g, dh, per * dh), self._init_net(g, dh, per * dh))

    def _init_net(self, g, din, dout):
        h = self.hidden
        def dense(a, b):   # stax.Dense: glorot normal weights, normal(1e-6) biases
            return (g.normal(0.0, np.sqrt(2.0 / (a + b)), size=(a, b)).astype(np.float32), g.normal(0.0, 1e-6, size=(b,)).astype(np.float32))
        return [dense(din, h), (), dense(h, h), (), dense(h, dout)]

    def __call__(self, rng, dim, **kwargs):
        from . import as_generator
        if dim % 2:
            raise ValueError("NeuralSplineCoupling needs an even number of dimensions")
        params = self.init_params(rng, dim)
        K, tail, hidden = self.K, self.B, self.hidden

        def run(params, x, inverse):
            import ctypes
            torch = __import__("torch")
            from .. import _lib
            from ..core import flatten_params
            L = _lib.lib()
            was_numpy = not hasattr(x, "detach")
            t = (torch.as_tensor(np.asarray(x, dtype=np.float32)) if was_numpy else x).to("cuda", dtype=torch.float32).contiguous()
            if t.dim() != 2 or t.shape[1] != dim:
                raise ValueError(f"expected inputs of shape [B, {dim}]")
            Bn = t.shape[0]
            flat = torch.as_tensor(flatten_params(params)).to(t.device)
            y = torch.empty_like(t)
            ld = torch.empty(Bn, device=t.device, dtype=torch.float32)
            ws = torch.empty(int(_lib.check(L.wf_nsc_workspace_bytes(Bn, dim, K), "wf_nsc_workspace_bytes")), device=t.device, dtype=torch.uint8)
            P = lambda a: ctypes.c_void_p(a.data_ptr())
            _lib.check(L.wf_nsc_fwd(P(t), Bn, dim, K, tail, hidden, P(flat), int(inverse), P(y), P(ld), P(ws), ws.numel(),
                                    ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)), "wf_nsc_fwd")
            return (y.cpu().numpy(), ld.cpu().numpy()) if was_numpy else (y, ld)

        def direct_fun(params, x, **kw):
            return run(params, x, False)

        def inverse_fun(params, z, **kw):
            return run(params, z, True)

        return params, direct_fun, inverse_fun
