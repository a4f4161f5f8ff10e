"""Oracle for the local-energy path -- TEST INFRASTRUCTURE ONLY.

A vectorised PyTorch (CPU, fp64 or fp32) restatement of Waveflow.psi (wavefunctions.py:54-71) in which the table lerp
carries the reference's custom derivative rule (isplines_jax.py:60-66, bsplines_jax.py:32-38: the derivative of the
order-nd lerp IS the order-(nd+1) lerp), so that torch.autograd reproduces what jax.hessian sees.  On top of it:
    laplacian            physics.py:50-52   (trace of the per-walker Hessian of psi)
    potential            physics.py:60-76   (soft-Coulomb, 1 space dimension)
    hamiltonian (H psi)  physics.py:79-93   (-0.5 * laplacian + V * psi)
The value path is checked against wf_oracle.c (tests/test_oracle_energy.py), which is pinned to the He golden grids; the
derivative path is parity-unpinned in the reference (the shipped energies come from a diverged run) and is validated by
finite differences of an exact-spline evaluation of the same model.
"""
import numpy as np
import torch

import oracle


class _Basis(torch.autograd.Function):
    """basis[..., j] = lerp(T[nd][j], x) with d/dx := the same lerp on table nd+1 (the reference's custom_jvp)."""

    @staticmethod
    def forward(ctx, x, tab, nd):
        n = tab.shape[-1] - 1
        xs = x * n
        il = torch.floor(xs).long()
        ir = torch.ceil(xs).long()
        ilc = torch.where(il < 0, il + tab.shape[-1], il).clamp(0, n)
        irc = torch.where(ir < 0, ir + tab.shape[-1], ir).clamp(0, n)
        T = tab[nd]                                  # [nb, n_mesh]
        yl = T[:, ilc].movedim(0, -1)                # [..., nb]
        yr = T[:, irc].movedim(0, -1)
        dx = x - il.to(x.dtype) / n
        ctx.save_for_backward(x, tab)
        ctx.nd = nd
        return yl + (yr - yl) * n * dx.unsqueeze(-1)

    @staticmethod
    def backward(ctx, g):
        x, tab = ctx.saved_tensors
        # the reference indexes cached_bases_dict[n_derivative + 1] with a traced integer (isplines_jax.py:65): JAX clamps an
        # out-of-range dynamic index, so the derivative of the highest cached order (nd = 3) is that same order again
        return (g * _Basis.apply(x, tab, min(ctx.nd + 1, tab.shape[0] - 1))).sum(-1), None, None


def table_spline(x, c, tab, nd=0):
    """sum_j c[..., j] * X_cached(x, j, nd)   (isplines_jax.py:78, bsplines_jax.py:44)"""
    return (c * _Basis.apply(x, tab, nd)).sum(-1)


class TorchWaveflow:
    """Waveflow (B-spline prior, IMADE layers, 'mean' or 'first' box) in torch; parameters = flat reference leaf order."""

    def __init__(self, D, n_layers, box, box_L, k, knots, i_reg, constr_left, dtype=torch.float64, n_mesh=2000, i_left=None, i_right=None,
                 p_left=None, p_right=None, i_gate=False, p_gate=False):
        """i_left / i_right / p_left / p_right: general {n_derivative: value} dictionaries (None: the zero-only defaults {0: 0} / {0: 1} and
        {0: 0} / {0: 0}, evaluated as a 0/1 mask); enforced literally, in dictionary order, as enforce_boundary_conditions does."""
        self.D, self.n_layers, self.box, self.L, self.k, self.i_reg, self.dtype = D, n_layers, box, float(box_L), k, float(i_reg), dtype
        self.i_gate, self.p_gate = bool(i_gate), bool(p_gate)   # set_nn_output_grad_to_zero (model_factory.py:64-67)
        self.i_bc = None if i_left is None and i_right is None else ({0: 0.0} if i_left is None else i_left, {0: 1.0} if i_right is None else i_right)
        self.p_bc = None if p_left is None and p_right is None else ({0: 0.0} if p_left is None else p_left, {0: 0.0} if p_right is None else p_right)
        self.constr_left = tuple(constr_left)
        I = oracle.table(oracle.KIND_I, k, knots, n_mesh)
        Bt, OB, b2o, o2b = oracle.ortho_b(k, knots, n_mesh)
        f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32)).to(dtype)      # the reference stores fp32 tables
        self.I, self.OB, self.Bplain, self.o2b = f32(I), f32(OB), f32(Bt), f32(o2b)
        self.i_nb, self.p_nb = I.shape[1], OB.shape[1]
        H = 64
        din = torch.arange(D)
        dh = torch.arange(H) % (D - 1)
        dout = torch.arange(D) - 1
        self.masks = [(dh[None, :] >= din[:, None]).to(dtype), (dh[None, :] >= dh[:, None]).to(dtype), (dout[None, :] >= dh[:, None]).to(dtype)]

    @staticmethod
    def _enforce(w, tab, left, right, is_I):
        """enforce_boundary_conditions before its final normalisation (isplines_jax.py:166-190, bsplines_jax.py:176-189): coefficient nd
        (left) / nb-1-nd (right) <- (value - sum_{j<nd} T^(nd)_j(end) c_j) / T^(nd)_nd(end).  w: [..., nb]; tab: [4][nb][n_mesh]."""
        nb = w.shape[-1]
        cols = list(w.unbind(-1))
        for nd, val in left.items():
            acc = sum(tab[nd, j, 0] * cols[j] for j in range(nd)) if nd else 0.0
            cols[nd] = (val - acc) / tab[nd, nd, 0] + 0 * cols[nd]
        for nd, val in right.items():
            if is_I and nd == 0:
                cols[nb - 1] = 0 * cols[nb - 1]
                continue
            acc = sum(tab[nd, nb - 1 - j, -1] * cols[nb - 1 - j] for j in range(nd)) if nd else 0.0
            cols[nb - nd - 1] = (val - acc) / tab[nd, nb - nd - 1, -1] + 0 * cols[nb - nd - 1]
        return torch.stack(cols, -1)

    def _net(self, p, off, n_out):
        D, H = self.D, 64
        sizes = [D * H, H, H * H, H, H * n_out * D, n_out * D, D * n_out]
        parts, o = [], off
        for s in sizes:
            parts.append(p[o:o + s]); o += s
        W0, b0, W1, b1, W2, b2 = parts[0].view(D, H), parts[1], parts[2].view(H, H), parts[3], parts[4].view(H, n_out * D), parts[5]
        return (W0, b0, W1, b1, W2, b2, parts[6].view(D, n_out)), o

    def _conditioner(self, net, x, n_out, sigmoid, gate=False):
        W0, b0, W1, b1, W2, b2, zero = net
        m0, m1, m2 = self.masks
        h = torch.tanh(x @ (W0 * m0) + b0)
        h = torch.tanh(h @ (W1 * m1) + b1)
        o = h @ (W2 * m2.repeat(1, n_out)) + b2                       # columns c = j * D + d
        p = o.view(-1, n_out, self.D).transpose(1, 2)                  # [B, D, n_out]
        if sigmoid:
            p = torch.sigmoid(p)
            zero = zero.abs()
        if gate:   # cubed_input_product = roll(cumprod(x^3), 1) with entry 0 set to 1 (model_factory.py:65-67)
            g = torch.cat([torch.ones_like(x[:, :1]), torch.cumprod(x ** 3, -1)[:, :-1]], -1)
            p = g[..., None] * p + zero
        return p / p.sum(-1, keepdim=True)

    def psi(self, flat, x):
        return self._eval(flat, x, False)

    def log_pdf(self, flat, x):
        """wavefunctions.py:33-52: sum_d log(psi_d^2 [/ 2 on constrained dimensions] + 1e-7) + log det"""
        return self._eval(flat, x, True)

    def _eval(self, flat, x, want_log_pdf):
        dt = self.dtype
        p = flat.to(dt) if torch.is_tensor(flat) else torch.as_tensor(np.asarray(flat, dtype=np.float32)).to(dt)
        D, L, tol, k = self.D, self.L, 1e-7, self.k
        # BoxTransformLayer
        if self.box == "mean":
            mean = x.mean(-1)
            l = mean - x[:, 0]
            w = x[:, -1] - x[:, 0]
            cols, space, ld = [], 2 * L, torch.zeros(x.shape[0], dtype=dt)
            for i in range(D - 1):
                diff = x[:, i + 1] - x[:, i]
                cols.append(diff / (space + tol))
                ld = ld - torch.log(torch.as_tensor(space + tol, dtype=dt) if not torch.is_tensor(space) else space + tol)
                space = space - diff
            cols.append((mean + L - l) / (2 * L - w + tol))
            ld = ld - torch.log(2 * L - w + tol)
            u = torch.stack(cols, -1)
        else:
            cols = [(x[:, 0] + L) / (2 * L)]
            for i in range(1, D):
                cols.append((x[:, i] - x[:, i - 1]) / (L - x[:, i - 1] + tol))
            u = torch.stack(cols, -1)
            ld = -np.log(2 * L) - torch.log(L - x[:, :-1] + tol).sum(-1)
        off = 0
        nb = self.i_nb
        for _ in range(self.n_layers):
            net, off = self._net(p, off, nb)
            w = self._conditioner(net, u, nb, True, self.i_gate) + self.i_reg
            # remove_bias (isplines_jax.py:196-202)
            scale = torch.ones(nb, dtype=dt)
            for i in range(k):
                scale[i + 1] = scale[i + 1] * (i + 1) / k
                scale[nb - (i + 2)] = scale[nb - (i + 2)] * (i + 1) / k
            w = w * scale
            w = w / w.sum(-1, keepdim=True)
            # enforce_boundary_conditions {0:0} / {0:1}
            if self.i_bc is None:
                keep = torch.ones(nb, dtype=dt); keep[0] = 0; keep[-1] = 0
                w = w * keep
            else:
                w = self._enforce(w, self.I, self.i_bc[0], self.i_bc[1], True)
            w = w / w.sum(-1, keepdim=True)
            y = table_spline(u, w, self.I, 0)
            dy = table_spline(u, w, self.I, 1)
            ld = ld + torch.log(dy + 1e-7).sum(-1)
            u = y.flip(-1)
        nbp = self.p_nb
        net, off = self._net(p, off, nbp)
        w = self._conditioner(net, u, nbp, False, self.p_gate)
        if self.p_bc is None:
            keep = torch.ones(nbp, dtype=dt); keep[0] = 0; keep[-1] = 0
            w = w * keep
        else:
            w = self._enforce(w, self.Bplain, self.p_bc[0], self.p_bc[1], False)
        w = w / torch.sqrt((w ** 2).sum(-1, keepdim=True))
        c = w @ self.o2b
        c = c / torch.sqrt((c ** 2).sum(-1, keepdim=True))
        uc = u.clamp(0.0, 1.0)
        ps = table_spline(uc, c, self.OB, 0)
        scale = torch.ones(D, dtype=dt)
        for d in self.constr_left:
            scale[d] = 1 / np.sqrt(2.0)
        if want_log_pdf:
            return torch.log(ps ** 2 * scale ** 2 + 1e-7).sum(-1) + ld
        return (ps * scale).prod(-1) * torch.exp(0.5 * ld)


def potential(x, protons):
    """physics.get_potential (physics.py:60-76), one space dimension: x [B, D], protons [P]."""
    pe = -(1 / torch.sqrt(1 + (protons[None, :, None] - x[:, None, :]) ** 2)).sum((-1, -2))
    D = x.shape[1]
    ee = torch.zeros(x.shape[0], dtype=x.dtype)
    for i in range(D):
        for j in range(i):
            ee = ee + 1 / torch.sqrt(1 + (x[:, i] - x[:, j]) ** 2)
    return pe + ee


def hamiltonian(model, flat, x, protons):
    """-> (H psi [B], psi [B], laplacian [B]) with the reference's autodiff semantics (physics.py:50-52, 79-93)."""
    x = torch.as_tensor(np.asarray(x), dtype=model.dtype).clone().requires_grad_(True)
    ps = model.psi(flat, x)
    (g,) = torch.autograd.grad(ps.sum(), x, create_graph=True)
    lap = torch.zeros_like(ps)
    for i in range(x.shape[1]):
        (gi,) = torch.autograd.grad(g[:, i].sum(), x, retain_graph=True)
        lap = lap + gi[:, i]
    V = potential(x.detach(), torch.as_tensor(np.asarray(protons, dtype=np.float64).reshape(-1), dtype=model.dtype))
    hpsi = -0.5 * lap + V * ps.detach()
    return hpsi.detach().numpy(), ps.detach().numpy(), lap.detach().numpy()


def _psi_lap(model, p, x):
    ps = model.psi(p, x)
    (g,) = torch.autograd.grad(ps.sum(), x, create_graph=True)
    lap = torch.zeros_like(ps)
    for i in range(x.shape[1]):
        (gi,) = torch.autograd.grad(g[:, i].sum(), x, create_graph=True)
        lap = lap + gi[:, i]
    return ps, lap


def psi_vjp(model, flat, x, w_psi, w_lap):
    """d/dparams sum_b (w_psi[b] * psi_b + w_lap[b] * laplacian_b): what reverse mode through the Hessian trace gives the
    reference (value_and_grad over physics.laplacian, vqmc.py:215-221).  -> flat gradient [n_params] (fp64)."""
    p = torch.as_tensor(np.asarray(flat, dtype=np.float32)).to(model.dtype).clone().requires_grad_(True)
    x = torch.as_tensor(np.asarray(x), dtype=model.dtype).clone().requires_grad_(True)
    ps, lap = _psi_lap(model, p, x)
    F = (torch.as_tensor(np.asarray(w_psi), dtype=model.dtype) * ps).sum() + (torch.as_tensor(np.asarray(w_lap), dtype=model.dtype) * lap).sum()
    (g,) = torch.autograd.grad(F, p)
    return g.detach().numpy()


def logpdf_vjp(model, flat, x, w):
    """d/dparams sum_b w[b] log_pdf_b -> flat gradient [n_params] (fp64)."""
    p = torch.as_tensor(np.asarray(flat, dtype=np.float32)).to(model.dtype).clone().requires_grad_(True)
    xt = torch.as_tensor(np.asarray(x), dtype=model.dtype)
    (g,) = torch.autograd.grad((torch.as_tensor(np.asarray(w), dtype=model.dtype) * model.log_pdf(p, xt)).sum(), p)
    return g.detach().numpy()


def vqmc_loss_grad(model, flat, x, protons, running_average):
    """loss_fn_efficient + its custom JVP (vqmc.py:193-212) -> (loss, gradient [n_params], local energies [B])."""
    p = torch.as_tensor(np.asarray(flat, dtype=np.float32)).to(model.dtype).clone().requires_grad_(True)
    xt = torch.as_tensor(np.asarray(x), dtype=model.dtype).clone().requires_grad_(True)
    ps, lap = _psi_lap(model, p, xt)
    V = potential(xt.detach(), torch.as_tensor(np.asarray(protons, dtype=np.float64).reshape(-1), dtype=model.dtype))
    hpsi = -0.5 * lap + V * ps
    e_loc = (hpsi / (ps + 1e-8)).detach()
    psd, hd = ps.detach(), hpsi.detach()
    # tangent rule: 2 t_psi (E - avg) / psi + (t_E psi - E t_psi) / psi^2, averaged over the batch
    a = (2.0 * (e_loc - running_average) / psd - hd / psd ** 2) / x.shape[0]
    c = (1.0 / psd) / x.shape[0]
    (g,) = torch.autograd.grad((a * ps).sum() + (c * hpsi).sum(), p)
    return float(e_loc.mean()), g.detach().numpy(), e_loc.numpy()


def _hpsi(model, p, x, protons):
    xt = torch.as_tensor(np.asarray(x), dtype=model.dtype).clone().requires_grad_(True)
    ps, lap = _psi_lap(model, p, xt)
    V = potential(xt.detach(), torch.as_tensor(np.asarray(protons, dtype=np.float64).reshape(-1), dtype=model.dtype))
    return ps, -0.5 * lap + V * ps, xt


def uniform_loss_grad(model, flat, x, protons):
    """value_and_grad of loss_fn_uniform (vqmc.py:143-154): mean(psi * H psi) / stop_gradient(mean(psi^2)), plain autograd."""
    p = torch.as_tensor(np.asarray(flat, dtype=np.float32)).to(model.dtype).clone().requires_grad_(True)
    ps, hpsi, _ = _hpsi(model, p, x, protons)
    loss = (ps * hpsi).mean() / (ps ** 2).mean().detach()
    (g,) = torch.autograd.grad(loss, p)
    return float(loss.detach()), g.detach().numpy()


def train_step_gradients(model, flat, x, protons, running_average, clip=10.0):
    """The gradient vqmc.train_step hands to the optimiser (vqmc.py:169-187), plain autograd of the reference's expressions:
    grad of mean(H psi / psi)  +  mean_b[ d log_pdf_b * (E_b / psi_b - running_average) ], clipped to [-10, 10];
    loss = mean(clip(E / psi, -100, 100)).  -> (gradient [n_params], loss)"""
    p = torch.as_tensor(np.asarray(flat, dtype=np.float32)).to(model.dtype).clone().requires_grad_(True)
    ps, hpsi, xt = _hpsi(model, p, x, protons)
    ne = hpsi / ps
    (ge,) = torch.autograd.grad(ne.mean(), p)
    w = ((ne - running_average) / len(ne)).detach()
    (gp,) = torch.autograd.grad((w * model.log_pdf(p, xt.detach())).sum(), p)
    loss = float(torch.clamp(ne.detach(), -100, 100).mean())
    g = ge + gp
    return (torch.clamp(g, -clip, clip) if clip is not None else g).detach().numpy(), loss


def he_model(dtype=torch.float64):
    return TorchWaveflow(2, 3, "mean", 10.0, 6, 23, 0.05, (0,), dtype=dtype)
