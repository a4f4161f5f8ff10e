"""RQS bijector: HIP kernel vs the oracle restatement and self-consistency.  Parity unpinned."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _params(N, K, n_deriv, seed):
    g = np.random.default_rng(seed)
    return (g.normal(size=(N, K)).astype(np.float32), g.normal(size=(N, K)).astype(np.float32),
            g.normal(size=(N, n_deriv)).astype(np.float32))


@pytest.mark.parametrize("K", [4, 6, 8, 32, 40])   # 6 and 40 take the generic (LDS-staged) kernel
@pytest.mark.parametrize("N", [1, 255, 1000, 4099])
def test_unconstrained_rqs_vs_oracle(K, N):
    from waveflow_amd.flows import unconstrained_RQS
    uw, uh, ud = _params(N, K, K - 1, K * 1000 + N)
    x = np.random.default_rng(1).uniform(-1.3, 1.3, size=N).astype(np.float32)
    x[:3] = [-1.0, 1.0, 0.0][:min(3, N)]
    y, ld, b = unconstrained_RQS(x, uw, uh, ud, tail_bound=1.0, return_bin_idx=True)
    yo, ldo, bo = oracle.rqs_batch(x, uw, uh, ud, left=-1.0, right=1.0, bottom=-1.0, top=1.0)
    tails = np.abs(x) > 1.0
    assert np.array_equal(y[tails], x[tails]) and np.all(ld[tails] == 0) and np.all(b[tails] == -1)
    # bin index: exact except where x sits within rounding of a knot (expf differs between libm and ocml)
    assert (b != bo).mean() < 2e-3
    same = b == bo
    # fp32 cumsum over K softmax terms with two expf implementations: a few ulp of the [-1, 1] range
    np.testing.assert_allclose(y[same], yo[same], rtol=0, atol=1e-5)
    np.testing.assert_allclose(ld[same], ldo[same], rtol=1e-4, atol=2e-4)
    # a flipped bin at a knot still gives a continuous map
    np.testing.assert_allclose(y, yo, rtol=0, atol=2e-5)


@pytest.mark.parametrize("K", [4, 8, 32])
def test_rqs_bin_search_is_bit_exact_on_injected_knots(K):
    """searchsorted (neural_splines.py:11-13: sum(x >= knots) - 1, eps on the last knot) in isolation.  The mismatches the test above
    tolerates come from the KNOTS (two expf implementations in the soft-max), not from the search: with equal unnormalised widths the
    soft-max is exactly 1 / K in any implementation (exp(0) = 1, K a power of two), the knots of kernel and oracle are the same fp32
    numbers, and then every bin index must agree -- on the knots themselves, on their fp32 neighbours, at both ends of the interval
    and at the tail bound."""
    from waveflow_amd.flows import unconstrained_RQS
    ideal = (-1.0 + 2.0 * np.arange(K + 1) / K).astype(np.float32)           # the knots up to the rounding of the cumulative sum
    pts = [ideal]
    for _ in range(4):                                                        # +- 1 .. 4 ulp around every knot
        pts.append(np.nextafter(pts[-1], np.float32(2.0)))
    lo = ideal
    for _ in range(4):
        lo = np.nextafter(lo, np.float32(-2.0))
        pts.append(lo)
    x = np.concatenate(pts + [np.random.default_rng(5).uniform(-1.05, 1.05, size=50000).astype(np.float32),
                              np.array([-1.0, 1.0, np.nextafter(np.float32(1.0), np.float32(2.0)), np.nextafter(np.float32(-1.0), np.float32(-2.0))], np.float32)])
    N = x.size
    uw = np.zeros((N, K), np.float32)
    uh = np.random.default_rng(6).normal(size=(N, K)).astype(np.float32)     # heights and derivatives do not enter the forward search
    ud = np.random.default_rng(7).normal(size=(N, K - 1)).astype(np.float32)
    y, ld, b = unconstrained_RQS(x, uw, uh, ud, tail_bound=1.0, return_bin_idx=True)
    yo, ldo, bo = oracle.rqs_batch(x, uw, uh, ud, left=-1.0, right=1.0, bottom=-1.0, top=1.0)
    assert np.array_equal(b, bo), (int((b != bo).sum()), x[b != bo][:8], b[b != bo][:8], bo[b != bo][:8])
    inside = np.abs(x) <= 1.0
    assert b[inside].min() == 0 and b[inside].max() == K - 1 and np.all(b[~inside] == -1)
    # every bin is hit, and the knots themselves belong to the bin on their right (x >= knot), the last one to the last bin (eps)
    assert np.array_equal(np.unique(b[inside]), np.arange(K))


@pytest.mark.parametrize("K", [8, 32])
def test_rqs_inverse_roundtrip_and_explicit_derivatives(K):
    from waveflow_amd.flows import RQS
    N = 20000
    uw, uh, ud = _params(N, K, K + 1, 7)
    x = np.random.default_rng(2).uniform(0.001, 0.999, size=N).astype(np.float32)
    y, ld = RQS(x, uw, uh, ud)
    yo, ldo, _ = oracle.rqs_batch(x, uw, uh, ud)
    np.testing.assert_allclose(y, yo, rtol=0, atol=1e-5)
    x2, ld2 = RQS(y, uw, uh, ud, inverse=True)
    ok = np.exp(ld) > 1e-2   # well-conditioned bins
    assert np.abs(x2 - x)[ok].max() < 2e-4
    assert np.abs(ld + ld2)[ok].max() < 2e-2
    assert np.all(y >= 0) and np.all(y <= 1) and np.all(np.diff(np.sort(x)) >= 0)
    # monotone: a sorted scan through one element's spline
    t = np.linspace(0, 1, 4001, dtype=np.float32)
    rep = lambda a: np.repeat(a[:1], t.size, axis=0)
    ys, lds = RQS(t, rep(uw), rep(uh), rep(ud))
    assert np.all(np.diff(ys) > -1e-6) and abs(ys[0]) < 1e-6 and abs(ys[-1] - 1) < 1e-5
    fd = np.diff(ys.astype(np.float64)) / np.diff(t.astype(np.float64))
    mid = np.exp(0.5 * (lds[1:] + lds[:-1]).astype(np.float64))
    assert np.median(np.abs(fd / mid - 1)) < 2e-3


def test_rqs_torch_tensors_and_errors():
    import torch
    from waveflow_amd import _lib
    from waveflow_amd.flows import unconstrained_RQS
    uw, uh, ud = (torch.from_numpy(a).cuda() for a in _params(512, 8, 7, 3))
    x = torch.rand(512, device="cuda") * 2 - 1
    y, ld = unconstrained_RQS(x, uw, uh, ud)
    assert y.is_cuda and y.shape == x.shape and torch.isfinite(y).all() and torch.isfinite(ld).all()
    with pytest.raises(ValueError):
        unconstrained_RQS(x, uw, uh, torch.zeros(512, 9, device="cuda"))
    L = _lib.lib()
    assert L.wf_rqs_fwd(None, None, None, None, 0, 2000, 1999, 0, 0.0, 1.0, 0.0, 1.0, None, None, None, None) == -1   # K too large
    assert L.wf_rqs_fwd(None, None, None, None, 10, 8, 7, 0, 0.0, 1.0, 0.0, 1.0, None, None, None, None) == -1       # null buffers


def _nsc_oracle(params, x, K, tail, inverse):
    """NumPy restatement of NeuralSplineCoupling (neural_splines.py:244-300) on top of the C oracle's RQS (oracle.rqs_batch)."""
    import oracle

    def fcnn(p, v):
        (W1, b1), _, (W2, b2), _, (W3, b3) = p
        h = np.tanh(v.astype(np.float64) @ W1 + b1)
        h = np.tanh(h @ W2 + b2)
        return h @ W3 + b3

    def half(p, cond, trans):
        dh = cond.shape[1]
        out = fcnn(p, cond).reshape(-1, dh, 3 * K - 1)
        W, H, D = out[..., :K], out[..., K:2 * K], out[..., 2 * K:]
        sm = lambda a: np.exp(a - a.max(-1, keepdims=True)) / np.exp(a - a.max(-1, keepdims=True)).sum(-1, keepdims=True)
        W, H = 2 * tail * sm(W), 2 * tail * sm(H)
        D = np.log1p(np.exp(D))
        y, ld, _ = oracle.rqs_batch(trans.reshape(-1).astype(np.float32), W.reshape(-1, K).astype(np.float32), H.reshape(-1, K).astype(np.float32),
                                 D.reshape(-1, K - 1).astype(np.float32), inverse=inverse, left=-tail, right=tail, bottom=-tail, top=tail)
        return y.reshape(trans.shape), ld.reshape(trans.shape).sum(1)

    dh = x.shape[1] // 2
    lower, upper = x[:, :dh], x[:, dh:]
    f1, f2 = params
    if not inverse:
        upper, l1 = half(f1, lower, upper)
        lower, l2 = half(f2, upper, lower)
    else:
        lower, l1 = half(f2, upper, lower)
        upper, l2 = half(f1, lower, upper)
    return np.concatenate([lower, upper], 1), l1 + l2


@pytest.mark.parametrize("dim,K,hidden", [(2, 5, 8), (4, 8, 8), (6, 5, 16)])
def test_neural_spline_coupling_layer(dim, K, hidden):
    """flows.NeuralSplineCoupling (the reference's tests/test_bijections.py:138 checks invertibility only): vs the NumPy / C
    restatement, inverse(direct(x)) = x, log-dets cancel."""
    from waveflow_amd import flows
    g = np.random.default_rng(dim)
    init_fun = flows.NeuralSplineCoupling(K=K, B=3, hidden_dim=hidden)
    params, direct_fun, inverse_fun = init_fun(11, dim)
    # the initial Dense weights are tiny: scale them so that the splines are far from the identity
    params = tuple([tuple(a * (1.5 if a.ndim == 2 else 3e4) for a in l) if l else () for l in net] for net in params)
    x = g.uniform(-3.5, 3.5, size=(4097, dim)).astype(np.float32)     # some points outside the +-3 tails (identity there)
    y, ld = direct_fun(params, x)
    yo, ldo = _nsc_oracle(params, x, K, 3.0, False)
    # (the restatement runs the conditioner in fp64, the kernel in fp32)
    assert np.abs(y - yo).max() < 5e-3 and np.quantile(np.abs(y - yo), 0.999) < 5e-4 and np.abs(ld - ldo).max() < 2e-2
    assert np.median(np.abs(y - yo)) < 2e-6
    assert np.abs(y - x).max() > 0.1                                   # it does transform
    xb, ldb = inverse_fun(params, y)
    # the inverse amplifies fp32 rounding by 1 / (local slope): judge the round trip by quantiles, and per point by the slope
    err = np.abs(xb - x).max(1)
    assert np.median(err) < 3e-4 and np.quantile(err, 0.9) < 1e-2
    assert np.median(np.abs(ld + ldb)) < 1e-3
    xo, _ = _nsc_oracle(params, y, K, 3.0, True)
    assert np.median(np.abs(xb - xo)) < 3e-4 and np.quantile(np.abs(xb - xo), 0.9) < 1e-2
    import torch
    yt, ldt = direct_fun(params, torch.as_tensor(x).cuda())
    assert torch.is_tensor(yt) and np.array_equal(yt.cpu().numpy(), y)


@pytest.mark.parametrize("dim,K,hidden,reverse,prior", [(2, 5, 8, True, "normal"), (4, 8, 8, False, "normal"), (6, 5, 32, True, "uniform"),
                                                        (2, 12, 8, True, "normal")])
def test_neural_spline_coupling_stack_as_a_model(dim, K, hidden, reverse, prior):
    """Flow(Serial((NeuralSplineCoupling [, Reverse]) x 3), Normal | Uniform): layer_kind WF_LAYER_NSC, log_pdf of the whole stack from ONE
    kernel launch, vs the NumPy / C restatement chained layer by layer; flow, inverse and sampler round trips."""
    import torch
    from waveflow_amd import flows
    g = np.random.default_rng(100 + dim)
    layer = lambda: flows.NeuralSplineCoupling(K=K, B=3, hidden_dim=hidden)
    items = []
    for _ in range(3):
        items += [layer(), flows.Reverse()] if reverse else [layer()]
    pr = flows.Normal(-0.25) if prior == "normal" else flows.Uniform()
    init = flows.Flow(flows.Serial(*items), pr, prior_support=None if prior == "normal" else (0.0, 1.0))
    params, log_pdf, sample = init(7, dim)
    assert log_pdf.model.desc.layer_kind == 2 and log_pdf.model.n_params > 0
    scale = lambda net: [tuple(a * (1.5 if a.ndim == 2 else 3e4) for a in l) if l else () for l in net]
    params = [tuple(scale(net) for net in p) if p else () for p in params]
    x = g.uniform(-3.4, 3.4, size=(4099, dim)).astype(np.float32)
    lp, u = log_pdf(params, x, return_sample=True)
    # oracle: layer by layer
    z, ld = x.astype(np.float32), np.zeros(len(x))
    for p in params:
        if p:
            z, l = _nsc_oracle(p, z, K, 3.0, False)
            z = z.astype(np.float32)
            ld = ld + l
        else:
            z = z[:, ::-1]
    if prior == "normal":
        want = ld + (-0.5 * (np.log(2 * np.pi) + (z.astype(np.float64) - 0.25) ** 2)).sum(1)
        zc = z
    else:
        want, zc = ld, np.clip(z, 0.0, 1.0)
    assert np.quantile(np.abs(u - zc), 0.999) < 2e-3 and np.median(np.abs(u - zc)) < 5e-6
    err = np.abs(lp - want)
    # (measured, scratch/nsc_err_diag.py, profiles/r03_nsc_error_vs_conditioning.txt: median 2 - 4e-6, 99th percentile 0.5 - 1.1e-4, worst walker
    # 2.7 - 4.3e-4, i.e. within 3 x (99 %) / 30 x (worst) of what a ONE-ulp change of the input does to the restatement's own result: the
    # biases of this test model are scaled by 3e4, so a few walkers sit where the conditioner is steep)
    assert np.median(err) < 2e-5 and np.quantile(err, 0.99) < 1e-3 and err.max() < 2e-2, (np.median(err), np.quantile(err, 0.99), err.max())
    m = log_pdf.model
    # flow / inverse round trip (exact inverse: a coupling layer conditions on the half it does not change)
    uf, ldf = m.flow(x)
    assert np.array_equal(np.asarray(ldf) + 0 * 0, np.asarray(ldf)) and np.abs(np.asarray(ldf) - ld).max() < 2e-2
    xb = np.asarray(m.inverse(uf))
    e = np.abs(xb - x).max(1)
    assert np.median(e) < 5e-4 and np.quantile(e, 0.9) < 2e-2
    # sampler: x = inverse(z), z ~ prior; log_pdf's latent returns z
    s, lat = sample(3, params, 2000, return_original_samples=True)
    s, lat = (v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v) for v in (s, lat))
    assert np.isfinite(s).all() and s.shape == (2000, dim)
    if prior == "normal":
        assert abs(lat.mean()) < 0.08 and abs(lat.std() - 1) < 0.08
    _, u2 = log_pdf(params, s, return_sample=True)
    d = np.abs(np.asarray(u2) - (lat if prior == "normal" else np.clip(lat, 0, 1))).max(1)
    assert np.median(d) < 1e-3 and np.quantile(d, 0.9) < 2e-2     # (three layers: three times the one-layer round-trip bound above)
    # what this model kind does not do fails loudly
    from waveflow_amd import _lib
    with pytest.raises(_lib.WfError):
        m.logpdf_vjp(x[:8], np.ones(8, np.float32))
    with pytest.raises(_lib.WfError):
        m.set_kernel("mfma")


def test_neural_spline_coupling_one_kernel_equals_staged_path(monkeypatch):
    """The bare layer (wf_nsc_fwd) takes the one-kernel stack for the built shapes; WF_NSC_STAGED=1 keeps the launch-per-half-step path:
    same function, same arithmetic for the spline, so the two agree to fp32 rounding of the conditioner's sums."""
    from waveflow_amd import flows
    g = np.random.default_rng(5)
    init_fun = flows.NeuralSplineCoupling(K=5, B=3, hidden_dim=8)
    params, direct_fun, inverse_fun = init_fun(11, 4)
    params = tuple([tuple(a * (1.5 if a.ndim == 2 else 3e4) for a in l) if l else () for l in net] for net in params)
    x = g.uniform(-3.5, 3.5, size=(5000, 4)).astype(np.float32)
    y1, l1 = direct_fun(params, x)
    monkeypatch.setenv("WF_NSC_STAGED", "1")
    y2, l2 = direct_fun(params, x)
    assert np.quantile(np.abs(y1 - y2), 0.999) < 1e-4 and np.quantile(np.abs(l1 - l2), 0.999) < 1e-3
