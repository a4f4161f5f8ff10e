"""Host-side handle on a wf_model (include/waveflow_hip.h) plus pytree helpers.

PyTorch is used only as plumbing: device buffers, the current HIP stream, torch.distributed.
"""
import ctypes
import os

import numpy as np

from . import _lib


def tree_leaves(tree, out=None):
    """Leaves in JAX pytree order for nested tuples / lists (depth first, left to right)."""
    if out is None:
        out = []
    if isinstance(tree, (tuple, list)):
        for t in tree:
            tree_leaves(t, out)
    elif tree is not None:
        out.append(tree)
    return out


class DeviceParams:
    """Parameters that live on the GPU: the flat float32 vector (reference leaf order) plus the pytree they unflatten to.
    Accepted wherever the closures take `params`; `tree()` downloads them as the reference's pytree of numpy arrays."""

    def __init__(self, template, flat, version=0):
        self.template, self.flat, self.version = template, flat, version

    def tree(self):
        from .checkpoint import unflatten_like
        return unflatten_like(self.template, self.flat.detach().cpu().numpy())


def flatten_params(tree):
    """-> contiguous float32 host vector in leaf order."""
    if isinstance(tree, DeviceParams):
        return np.ascontiguousarray(tree.flat.detach().cpu().numpy(), dtype=np.float32)
    leaves = tree_leaves(tree)
    parts = []
    for a in leaves:
        if hasattr(a, "detach"):  # torch tensor
            a = a.detach().to("cpu").numpy()
        parts.append(np.asarray(a, dtype=np.float32).reshape(-1))
    if not parts:
        return np.zeros(0, np.float32)
    return np.ascontiguousarray(np.concatenate(parts))


def _torch():
    import torch
    return torch


class DeviceModel:
    """Owns one wf_model on one GPU."""

    def __init__(self, desc, device=None):
        torch = _torch()
        L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.WfError(_lib.ERR_NO_DEVICE, "wf_model_create")
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        self.desc = desc
        h = ctypes.c_void_p()
        _lib.check(L.wf_model_create(ctypes.byref(desc), self.device, ctypes.byref(h)), "wf_model_create")
        self._h = h
        self.n_params = int(L.wf_model_param_count(h))
        self.i_nb = int(L.wf_model_n_bases(h, 0))
        self.p_nb = int(L.wf_model_n_bases(h, 1))
        self._flat = None
        self._vjp_ws = None
        self._dev_key = None
        self.D = int(desc.n_dim)
        self.n_layers = int(desc.n_flow_layers)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().wf_model_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- parameters
    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32).reshape(-1)
        if flat.size != self.n_params:
            raise ValueError(f"expected {self.n_params} parameters, got {flat.size}")
        _lib.check(_lib.lib().wf_model_set_params(self._h, flat.ctypes.data, flat.size, self._stream()), "wf_model_set_params")
        self._flat = flat.copy()

    def set_params_device(self, flat_dev):
        """Parameters from a float32 cuda vector: asynchronous, no host copy (wf_model_set_params_device)."""
        if flat_dev.numel() != self.n_params or str(flat_dev.dtype) != "torch.float32" or not flat_dev.is_cuda:
            raise ValueError(f"expected a float32 cuda vector of {self.n_params} parameters")
        flat_dev = flat_dev.contiguous()
        _lib.check(_lib.lib().wf_model_set_params_device(self._h, self._p(flat_dev), flat_dev.numel(), self._stream()),
                   "wf_model_set_params_device")
        self._flat = None

    def ensure_params(self, tree):
        """Upload `tree` unless it equals what the device already holds."""
        if isinstance(tree, DeviceParams):
            key = (id(tree.flat), tree.version)
            if self._dev_key != key:
                self.set_params_device(tree.flat)
                self._dev_key = key
            return
        self._dev_key = None
        flat = flatten_params(tree)
        if self._flat is None or flat.size != self._flat.size or not np.array_equal(flat, self._flat):
            self.set_params(flat)

    def set_kernel(self, kind):
        kind = {"auto": _lib.KERNEL_AUTO, "scalar": _lib.KERNEL_SCALAR, "mfma": _lib.KERNEL_MFMA, "wave": _lib.KERNEL_WAVE}.get(kind, kind)
        _lib.check(_lib.lib().wf_model_set_kernel(self._h, int(kind)), "wf_model_set_kernel")

    # ---- plumbing
    def _stream(self):
        torch = _torch()
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _to_dev(self, x):
        """-> (float32 contiguous cuda tensor [B, D], converter for outputs)"""
        torch = _torch()
        was_numpy = not hasattr(x, "detach")
        t = torch.as_tensor(np.asarray(x, dtype=np.float32)) if was_numpy else x
        squeeze = t.dim() == 1  # wavefunctions.py:35-36 promotes a single walker to [1, D]
        if squeeze:
            t = t[None]
        if t.dim() != 2 or t.shape[1] != self.D:
            raise ValueError(f"expected inputs of shape [B, {self.D}], got {tuple(t.shape)}")
        t = t.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        back = (lambda o: o.cpu().numpy()) if was_numpy else (lambda o: o)
        return t, back

    def _new(self, shape, dtype=None):
        torch = _torch()
        return torch.empty(shape, device=f"cuda:{self.device}", dtype=dtype or torch.float32)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None

    # ---- hot path
    def _eval(self, fn, x, return_sample, return_bin_idx):
        torch = _torch()
        t, back = self._to_dev(x)
        B = t.shape[0]
        out = self._new((B,))
        u = self._new((B, self.D)) if return_sample else None
        idx = self._new((B, self.n_layers + 1, self.D, 2), torch.int32) if return_bin_idx else None
        if idx is not None:
            idx.zero_()
        _lib.check(fn(self._h, self._p(t), B, self._p(out), self._p(u), self._p(idx), self._stream()), fn.__name__)
        res = [back(out)]
        if return_sample:
            res.append(back(u))
        if return_bin_idx:
            res.append(back(idx))
        return res[0] if len(res) == 1 else tuple(res)

    def log_pdf(self, x, return_sample=False, return_bin_idx=False):
        return self._eval(_lib.lib().wf_logpdf_fwd, x, return_sample, return_bin_idx)

    def psi(self, x, return_sample=False, return_bin_idx=False):
        return self._eval(_lib.lib().wf_psi_fwd, x, return_sample, return_bin_idx)

    def psi_antisym(self, x, return_inversions=False):
        """psi(sort(x)) * (-1)^inversions(x) for walkers in any particle order (helpers.py:55-58, coordinates.py:41-51): sort and sign on the device."""
        torch = _torch()
        t, back = self._to_dev(x)
        B = t.shape[0]
        out = self._new((B,))
        inv = self._new((B,), torch.int32) if return_inversions else None
        _lib.check(_lib.lib().wf_psi_antisym_fwd(self._h, self._p(t), B, self._p(out), self._p(inv), self._stream()), "wf_psi_antisym_fwd")
        return (back(out), back(inv)) if return_inversions else back(out)

    def log_pdf_unsorted(self, x):
        """log_pdf(sort(x)): rows in any particle order, sorted on the device."""
        t, back = self._to_dev(x)
        B = t.shape[0]
        out = self._new((B,))
        _lib.check(_lib.lib().wf_logpdf_unsorted_fwd(self._h, self._p(t), B, self._p(out), self._stream()), "wf_logpdf_unsorted_fwd")
        return back(out)

    def flow(self, x):
        t, back = self._to_dev(x)
        B = t.shape[0]
        u, ld = self._new((B, self.D)), self._new((B,))
        _lib.check(_lib.lib().wf_flow_fwd(self._h, self._p(t), B, self._p(u), self._p(ld), self._stream()), "wf_flow_fwd")
        return back(u), back(ld)

    def layer(self, l, u_in, return_bin_idx=False):
        torch = _torch()
        t, back = self._to_dev(u_in)
        B = t.shape[0]
        y, ld = self._new((B, self.D)), self._new((B,))
        idx = self._new((B, self.D, 2), torch.int32) if return_bin_idx else None
        if idx is not None:
            idx.zero_()
        _lib.check(_lib.lib().wf_layer_fwd(self._h, int(l), self._p(t), B, self._p(y), self._p(ld), self._p(idx), self._stream()),
                   "wf_layer_fwd")
        if return_bin_idx:
            return back(y), back(ld), back(idx)
        return back(y), back(ld)

    def inverse(self, u, exact=False):
        """Serial.inverse_fun; exact=False reproduces the reference's IMADE.inverse_fun (conditioner on its inputs)."""
        t, back = self._to_dev(u)
        B = t.shape[0]
        x = self._new((B, self.D))
        _lib.check(_lib.lib().wf_inverse_fwd(self._h, self._p(t), B, self._p(x), int(bool(exact)), self._stream()), "wf_inverse_fwd")
        return back(x)

    def sample(self, seed, num_samples, return_latent=False, exact=False):
        """-> x [n, D] on the device (and the prior-space samples when return_latent)."""
        B = int(num_samples)
        x = self._new((B, self.D))
        lat = self._new((B, self.D)) if return_latent else None
        _lib.check(_lib.lib().wf_sample(self._h, ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), B, self._p(x), self._p(lat),
                                        int(bool(exact)), self._stream()), "wf_sample")
        return (x, lat) if return_latent else x

    def hamiltonian(self, x, protons, return_psi=False, return_laplacian=False):
        """H psi = -1/2 laplacian(psi) + V psi (physics.construct_hamiltonian_function); protons: 1-D positions."""
        t, back = self._to_dev(x)
        B = t.shape[0]
        pr = np.ascontiguousarray(np.asarray(protons, dtype=np.float32).reshape(-1))
        h = self._new((B,))
        ps = self._new((B,)) if return_psi else None
        lap = self._new((B,)) if return_laplacian else None
        _lib.check(_lib.lib().wf_hamiltonian_fwd(self._h, self._p(t), B, pr.ctypes.data if pr.size else None, pr.size, self._p(h),
                                                 self._p(ps), self._p(lap), self._stream()), "wf_hamiltonian_fwd")
        res = [back(h)]
        if return_psi:
            res.append(back(ps))
        if return_laplacian:
            res.append(back(lap))
        return res[0] if len(res) == 1 else tuple(res)

    def psi_vjp(self, x, w_psi, w_lap):
        """grad[p] = sum_b (w_psi[b] d psi_b/d theta_p + w_lap[b] d laplacian_b/d theta_p) -> torch.cuda float32 [n_params]."""
        torch = _torch()
        L = _lib.lib()
        t, _ = self._to_dev(x)
        B = t.shape[0]
        wp = torch.as_tensor(w_psi, dtype=torch.float32).to(t.device).contiguous()
        wl = torch.as_tensor(w_lap, dtype=torch.float32).to(t.device).contiguous()
        if wp.numel() != B or wl.numel() != B:
            raise ValueError("w_psi / w_lap must have one entry per walker")
        nbytes = _lib.check(L.wf_psi_vjp_workspace_bytes(self._h, B), "wf_psi_vjp_workspace_bytes")
        if self._vjp_ws is None or self._vjp_ws.numel() < nbytes:
            self._vjp_ws = self._workspace(nbytes, t.device)
        grad = self._new((self.n_params,))
        _lib.check(L.wf_psi_vjp(self._h, self._p(t), B, self._p(wp), self._p(wl), self._p(grad), self._p(self._vjp_ws),
                                self._vjp_ws.numel(), self._stream()), "wf_psi_vjp")
        return grad

    def logpdf_vjp(self, x, w):
        """grad[p] = sum_b w[b] d log_pdf_b / d theta_p -> torch.cuda float32 [n_params]."""
        torch = _torch()
        L = _lib.lib()
        t, _ = self._to_dev(x)
        B = t.shape[0]
        wt = torch.as_tensor(w, dtype=torch.float32).to(t.device).contiguous()
        if wt.numel() != B:
            raise ValueError("w must have one entry per row of x")
        nbytes = _lib.check(L.wf_logpdf_vjp_workspace_bytes(self._h, B), "wf_logpdf_vjp_workspace_bytes")
        if self._vjp_ws is None or self._vjp_ws.numel() < nbytes:
            self._vjp_ws = self._workspace(nbytes, t.device)
        grad = self._new((self.n_params,))
        _lib.check(L.wf_logpdf_vjp(self._h, self._p(t), B, self._p(wt), self._p(grad), self._p(self._vjp_ws), self._vjp_ws.numel(),
                                   self._stream()), "wf_logpdf_vjp")
        return grad

    def logpdf_loss_grad(self, x, weight):
        """(log_pdf [B], weight * sum_b d log_pdf_b / d theta [n_params]) from one forward and one reverse sweep."""
        L = _lib.lib()
        t, _ = self._to_dev(x)
        B = t.shape[0]
        nbytes = _lib.check(L.wf_logpdf_vjp_workspace_bytes(self._h, B), "wf_logpdf_vjp_workspace_bytes")
        if self._vjp_ws is None or self._vjp_ws.numel() < nbytes:
            self._vjp_ws = self._workspace(nbytes, t.device)
        lp, grad = self._new((B,)), self._new((self.n_params,))
        _lib.check(L.wf_logpdf_loss_grad(self._h, self._p(t), B, float(weight), self._p(lp), self._p(grad), self._p(self._vjp_ws),
                                         self._vjp_ws.numel(), self._stream()), "wf_logpdf_loss_grad")
        return lp, grad

    def vqmc_loss_grad(self, x, protons, running_average, global_count=None):
        """loss_fn_efficient and its gradient (vqmc.py:193-221) for the walkers x on this device, one fused pass.
        -> (sums fp64 [sum E_L, sum E_L^2, count] (torch.cuda), grad float32 [n_params] scaled by 1/global_count)."""
        torch = _torch()
        L = _lib.lib()
        t, _ = self._to_dev(x)
        B = t.shape[0]
        pr = np.ascontiguousarray(np.asarray(protons, dtype=np.float32).reshape(-1))
        nbytes = _lib.check(L.wf_psi_vjp_workspace_bytes(self._h, B), "wf_psi_vjp_workspace_bytes")
        if self._vjp_ws is None or self._vjp_ws.numel() < nbytes:
            self._vjp_ws = self._workspace(nbytes, t.device)
        el, grad = self._new((B,)), self._new((self.n_params,))
        inv = 1.0 / float(global_count if global_count else max(B, 1))
        _lib.check(L.wf_vqmc_loss_grad(self._h, self._p(t), B, pr.ctypes.data if pr.size else None, pr.size, float(running_average), inv,
                                       self._p(el), self._p(grad), self._p(self._vjp_ws), self._vjp_ws.numel(), self._stream()),
                   "wf_vqmc_loss_grad")
        return self.block_sums(el), grad

    def adam_step(self, x, g, m, v, step, step_size, b1=0.9, b2=0.999, eps=1e-8):
        """In-place Adam update of the cuda vectors x, m, v with the gradient g (wf_adam_step)."""
        for t in (x, g, m, v):
            if not t.is_cuda or not t.is_contiguous() or t.numel() != x.numel() or str(t.dtype) != "torch.float32":
                raise ValueError("adam_step needs contiguous float32 cuda vectors of equal length")
        _lib.check(_lib.lib().wf_adam_step(self._p(x), self._p(g), self._p(m), self._p(v), x.numel(), int(step), float(step_size), float(b1),
                                           float(b2), float(eps), self._stream()), "wf_adam_step")

    def make_train_state(self, x, m, v, first_step, ring_len=128, defer_eval_tables=False):
        """Device-side state of wf_vqmc_train_step around the Adam vectors x, m, v (float32 cuda, updated in place).
        defer_eval_tables: the steps skip the tables only the large-batch evaluation kernel reads; the caller refreshes them with
        set_params_device(x) before evaluating (ensure_params does so for a DeviceParams of a new version)."""
        torch = _torch()
        dev = x.device
        st = {"x": x, "m": m, "v": v, "ring_len": int(ring_len),
              "counter": torch.tensor([int(first_step)], dtype=torch.int64, device=dev),
              "running_average": torch.zeros(1, dtype=torch.float32, device=dev),
              "ring": torch.zeros(int(ring_len), 3, dtype=torch.float64, device=dev)}
        st["c"] = _lib.TrainState(x.data_ptr(), m.data_ptr(), v.data_ptr(), st["counter"].data_ptr(), st["running_average"].data_ptr(),
                                  st["ring"].data_ptr(), int(ring_len), int(bool(defer_eval_tables)))
        return st

    def train_step(self, st, seed, batch, protons, step_size, b1=0.9, b2=0.999, eps=1e-8, exact_sampler=False):
        """One whole training step on the device (wf_vqmc_train_step): no host work, capturable in a hipGraph."""
        L = _lib.lib()
        pr = np.ascontiguousarray(np.asarray(protons, dtype=np.float32).reshape(-1))
        nbytes = _lib.check(L.wf_vqmc_train_step_workspace_bytes(self._h, int(batch)), "wf_vqmc_train_step_workspace_bytes")
        if st.get("ws") is None or st["ws"].numel() < nbytes:
            st["ws"] = self._workspace(nbytes, st["x"].device)
        _lib.check(L.wf_vqmc_train_step(self._h, ctypes.byref(st["c"]), int(seed), int(batch), pr.ctypes.data if pr.size else None, pr.size,
                                        float(step_size), float(b1), float(b2), float(eps), int(bool(exact_sampler)), self._p(st["ws"]),
                                        st["ws"].numel(), self._stream()), "wf_vqmc_train_step")
        self._flat = None
        self._dev_key = None

    def train_step_local(self, st, seed, batch_local, protons, inv_global_batch, red, exact_sampler=False):
        """First half of a sharded training step (wf_vqmc_train_step_local): this rank's walkers -> red[n_params + 3] (float64 cuda)
        = [gradient contribution, sum E_L, sum E_L^2, local walkers]; the caller all-reduces red and calls train_step_apply."""
        L = _lib.lib()
        pr = np.ascontiguousarray(np.asarray(protons, dtype=np.float32).reshape(-1))
        nbytes = _lib.check(L.wf_vqmc_train_step_workspace_bytes(self._h, int(batch_local)), "wf_vqmc_train_step_workspace_bytes")
        if st.get("ws") is None or st["ws"].numel() < nbytes:
            st["ws"] = self._workspace(nbytes, st["x"].device)
        if str(red.dtype) != "torch.float64" or red.numel() != self.n_params + 3 or not red.is_cuda or not red.is_contiguous():
            raise ValueError("red must be a contiguous float64 cuda vector of n_params + 3 entries")
        _lib.check(L.wf_vqmc_train_step_local(self._h, ctypes.byref(st["c"]), int(seed), int(batch_local), pr.ctypes.data if pr.size else None,
                                              pr.size, float(inv_global_batch), int(bool(exact_sampler)), self._p(red), self._p(st["ws"]),
                                              st["ws"].numel(), self._stream()), "wf_vqmc_train_step_local")

    def train_step_apply(self, st, red, step_size, b1=0.9, b2=0.999, eps=1e-8):
        """Second half (wf_vqmc_train_step_apply): Adam with the reduced gradient, image refill, loss ring, counter."""
        _lib.check(_lib.lib().wf_vqmc_train_step_apply(self._h, ctypes.byref(st["c"]), self._p(red), float(step_size), float(b1), float(b2),
                                                       float(eps), self._stream()), "wf_vqmc_train_step_apply")
        self._flat = None
        self._dev_key = None

    def mle_train_step(self, st, x, step_size, b1=0.9, b2=0.999, eps=1e-8):
        """One maximum-likelihood epoch on the device (wf_mle_train_step): no host work, capturable in a hipGraph."""
        L = _lib.lib()
        nbytes = _lib.check(L.wf_mle_train_step_workspace_bytes(self._h, int(x.shape[0])), "wf_mle_train_step_workspace_bytes")
        if st.get("ws") is None or st["ws"].numel() < nbytes:
            st["ws"] = self._workspace(nbytes, st["x"].device)
        _lib.check(L.wf_mle_train_step(self._h, ctypes.byref(st["c"]), self._p(x), int(x.shape[0]), float(step_size), float(b1), float(b2),
                                       float(eps), self._p(st["ws"]), st["ws"].numel(), self._stream()), "wf_mle_train_step")
        self._flat = None
        self._dev_key = None

    @staticmethod
    def _workspace(nbytes, device):
        torch = _torch()
        ws = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
        if os.environ.get("WF_POISON"):
            ws.fill_(0xFF)   # NaN patterns: see dev_alloc in wf_model.cpp
        return ws

    def block_sums(self, v):
        """fp64 [sum v, sum v^2, count] on the device (deterministic order)."""
        torch = _torch()
        L = _lib.lib()
        v = v.contiguous()
        ws = torch.empty(int(L.wf_block_sums_workspace_bytes(v.numel())), device=v.device, dtype=torch.uint8)
        out = torch.empty(3, device=v.device, dtype=torch.float64)
        _lib.check(L.wf_block_sums(self._p(v), v.numel(), self._p(out), self._p(ws), ws.numel(), self._stream()), "wf_block_sums")
        return out


def build_tables(kind, degree, n_internal_knots, n_mesh=2000):
    """Host-only: fp64 [4][n_bases][n_mesh] (+ (b_to_ob, ob_to_b) for kind == SPLINE_OB)."""
    L = _lib.lib()
    nb = _lib.check(L.wf_tables_build(kind, degree, n_internal_knots, n_mesh, None, None, None), "wf_tables_build")
    out = np.zeros((4, nb, n_mesh))
    if kind == _lib.SPLINE_OB:
        b2o, o2b = np.zeros((nb, nb)), np.zeros((nb, nb))
        _lib.check(L.wf_tables_build(kind, degree, n_internal_knots, n_mesh, out.ctypes.data, b2o.ctypes.data, o2b.ctypes.data),
                   "wf_tables_build")
        return out, b2o, o2b
    _lib.check(L.wf_tables_build(kind, degree, n_internal_knots, n_mesh, out.ctypes.data, None, None), "wf_tables_build")
    return out
