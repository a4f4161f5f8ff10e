"""Inverse / sampling direction on the HIP path vs the oracle.  PRNG parity is unpinned (the reference
draws with JAX's threefry); the deterministic inverse is compared value by value, the sampler distributionally."""
import numpy as np
import pytest

import oracle
from conftest import sorted_walkers

pytestmark = pytest.mark.gpu


def he(he_flat):
    from waveflow_amd import checkpoint, model_factory
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6,
                                                n_flow_layers=3, box_size=10, xu_coord_type="mean")
    params, psi, log_pdf, sample = init_fun(0, 2)
    params = checkpoint.unflatten_like(params, he_flat)
    log_pdf.model.ensure_params(params)
    return params, psi, log_pdf, sample, oracle.he_model(10.0)


@pytest.mark.parametrize("exact", [False, True])
def test_serial_inverse_vs_oracle(he_flat, exact):
    params, psi, log_pdf, sample, om = he(he_flat)
    g = np.random.default_rng(0)
    u = g.uniform(0.02, 0.98, size=(3000, 2)).astype(np.float32)
    x = log_pdf.model.inverse(u, exact=exact)
    xo = om.inverse(he_flat, u, exact=exact)
    d = np.abs(x - xo)
    # bisection: identical decisions except where spline(mid) - y changes sign within rounding => off by <= tol * slope
    assert np.median(d) < 1e-5 and (d > 2e-3).mean() < 5e-3, (np.median(d), d.max())
    if exact:
        # the true inverse: direct(inverse(u)) == u up to the bisection tolerance (1e-6 per layer, amplified by 1/slope)
        u2, _ = log_pdf.model.flow(x)
        assert np.median(np.abs(u2 - u)) < 2e-5 and np.abs(u2 - u).max() < 5e-3


def test_reference_inverse_quirk_is_reproduced(he_flat, golden):
    """made.py:88: with the conditioner evaluated on the inputs, inverse(direct(x)) != x for columns > 0 -- the reference's
    published MFlow reconstruction distances (2.4e-2 of the unit box) are of that size."""
    params, psi, log_pdf, sample, om = he(he_flat)
    sp = np.sort(golden["he_golden"]["sample_points"], -1).astype(np.float32)
    u, _ = log_pdf.model.flow(sp)
    x_ref = log_pdf.model.inverse(u, exact=False)
    x_exact = log_pdf.model.inverse(u, exact=True)
    assert np.abs(x_exact - sp).max() < 2e-3
    assert np.median(np.abs(x_ref - sp)) > 0.05     # the quirk: percent-level errors in box units
    np.testing.assert_allclose(x_ref, om.inverse(he_flat, u), atol=2e-3)


def test_sampler_draws_from_the_prior_columns_and_is_reproducible(he_flat):
    import torch
    params, psi, log_pdf, sample, om = he(he_flat)
    m = log_pdf.model
    n = 200000
    x, lat = m.sample(1234, n, return_latent=True, exact=True)
    x2 = m.sample(1234, n, exact=True)
    assert torch.equal(x, x2)
    assert not torch.equal(x, m.sample(1235, n, exact=True))
    lat = lat.cpu().numpy()
    assert lat.min() >= 0 and lat.max() <= 1
    # column 0: the conditioner sees zeros, its density is the same for every walker -> compare the histogram
    grid = np.linspace(0, 1, 2001)
    dens, ymax = om.prior_column_density(he_flat, [0, 0], 0, grid)
    assert dens.max() <= ymax * 1.0001
    cdf = np.concatenate([[0], np.cumsum(0.5 * (dens[1:] + dens[:-1]) * np.diff(grid))])
    cdf /= cdf[-1]
    emp = np.searchsorted(np.sort(lat[:, 0]), grid) / n
    assert np.abs(emp - cdf).max() < 4.0 / np.sqrt(n)       # Kolmogorov-Smirnov, ~1e-6 false-alarm level
    # column 1 given column 0 in a narrow bin
    sel = np.abs(lat[:, 0] - 0.7) < 0.004
    dens1, _ = om.prior_column_density(he_flat, [0.7, 0], 1, grid)
    cdf1 = np.concatenate([[0], np.cumsum(0.5 * (dens1[1:] + dens1[:-1]) * np.diff(grid))]); cdf1 /= cdf1[-1]
    emp1 = np.searchsorted(np.sort(lat[sel, 1]), grid) / sel.sum()
    assert np.abs(emp1 - cdf1).max() < 4.0 / np.sqrt(sel.sum()) + 0.02
    # exact inverse => x ~ |psi|^2: direct(x) gives back the latent, walkers are sorted and inside the box
    xn = x.cpu().numpy()
    assert np.all(xn[:, 0] <= xn[:, 1] + 1e-4) and np.abs(xn).max() <= 10.0 + 1e-3
    u, _ = m.flow(x)
    assert np.median(np.abs(u.cpu().numpy() - lat)) < 2e-5
    # <log psi^2> over |psi|^2 samples is larger than over uniform walkers (sanity of the whole chain)
    lp = log_pdf(params, x)
    assert lp.mean().item() > -3.0


def test_closure_surface_sample_and_inverse(he_flat):
    from waveflow_amd import flows, model_factory, flatten_params
    params, psi, log_pdf, sample, om = he(he_flat)
    s = sample(3, params, 250)
    assert tuple(s.shape) == (250, 2)
    s2, lat = sample(3, params, 250, return_original_samples=True)
    assert (s - s2).abs().max().item() == 0 and tuple(lat.shape) == (250, 2)
    # per-layer protocol: IMADE alone
    mt = model_factory.get_masked_transform
    p, dfun, ifun = flows.IMADE(mt(), 5, 16, 0.05, 1e-6)(0, 2)
    u = np.random.default_rng(0).uniform(0.05, 0.95, size=(500, 2)).astype(np.float32)
    y, ld = dfun(p, u)
    x_back, zero = ifun(p, y, exact=True)
    assert zero == 0 and np.abs(x_back - u).max() < 1e-4
    # Flow (affine MADE + Normal) and MFlow samplers run and invert exactly (MADE.inverse_fun is the true inverse)
    init = flows.Flow(flows.Serial(*(flows.MADE(mt(return_simple_masked_transform=True)), flows.Reverse()) * 2), flows.Normal(-0.5))
    fp, flp, fsample = init(3, 2)
    xs, lat = fsample(0, fp, 20000, return_original_samples=True)
    uu, _ = flp.model.flow(xs)
    assert (uu - lat).abs().max().item() < 1e-3
    assert abs(lat.mean().item()) < 0.03 and abs(lat.std().item() - 1) < 0.03
    init = model_factory.get_model(n_flow_layers=2, i_spline_reg=0.02)
    mp, mlp, msample = init(1, 2)
    xs = msample(0, mp, 1000)
    assert xs.min().item() >= 0 and xs.max().item() <= 1
    # BoxTransformLayer alone
    _, bdir, brev = flows.BoxTransformLayer(3.0, "first")(0, 3)
    xw = sorted_walkers(100, 3, 3.0, 1)
    ub, _ = bdir((), xw)
    np.testing.assert_allclose(brev((), ub)[0], xw, atol=1e-5)


def test_wave_and_one_lane_samplers_draw_from_the_same_distribution(he_flat, monkeypatch):
    """wf_sample / wf_inverse_fwd switch kernels at 2^17 walkers (WF_WAVE_SAMPLE_MAX moves the switch): both kernels produce the
    same marginals (two-sample Kolmogorov-Smirnov on each coordinate and on the latent columns), and inverse() agrees."""
    from scipy import stats
    params, psi, log_pdf, sample, om = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")                             # (not the staged sampler of large batches: its own test below)
    for exact in (True, False):
        monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", "100000000")                 # one wave per walker
        xa, la = m.sample(7, 30000, return_latent=True, exact=exact)
        monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", "0")                         # one lane per walker
        xb, lb = m.sample(8, 40000, return_latent=True, exact=exact)
        for c in range(2):
            for a, b in ((xa, xb), (la, lb)):
                p = stats.ks_2samp(a[:, c].cpu().numpy(), b[:, c].cpu().numpy()).pvalue
                assert p > 1e-4, (exact, c, p)
    u = np.random.default_rng(0).uniform(0.01, 0.99, size=(40000, 2)).astype(np.float32)
    big = m.inverse(u, exact=True)                                            # one lane per walker
    monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", "100000000")
    small = m.inverse(u, exact=True)                                          # one wave per walker
    d = np.abs(big - small)
    assert np.median(d) < 1e-6 and np.quantile(d, 0.999) < 2e-4 and d.max() < 2e-3


def test_wave_sampler_with_more_than_32_bases(monkeypatch):
    """33 knots at k = 6 (39 / 38 bases): the wave sampler in its 64-row layout against the one-lane kernel -- same marginals, same
    inverse; and direct(inverse(u)) = u."""
    from scipy import stats
    from waveflow_amd import model_factory
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=33,
                                                n_i_internal_knots=33, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6,
                                                n_flow_layers=2, box_size=10, xu_coord_type="mean")
    params, psi, log_pdf, sample = init_fun(3, 2)
    m = psi.model
    m.ensure_params(params)
    assert (m.i_nb, m.p_nb) == (39, 38)
    for exact in (True, False):
        monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", "100000000")                 # one wave per walker
        xa, la = m.sample(7, 30000, return_latent=True, exact=exact)
        monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", "0")                         # one lane per walker
        xb, lb = m.sample(8, 40000, return_latent=True, exact=exact)
        for c in range(2):
            for a, b in ((xa, xb), (la, lb)):
                p = stats.ks_2samp(a[:, c].cpu().numpy(), b[:, c].cpu().numpy()).pvalue
                assert p > 1e-4, (exact, c, p)
    u = np.random.default_rng(0).uniform(0.01, 0.99, size=(20000, 2)).astype(np.float32)
    big = m.inverse(u, exact=True)
    monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", "100000000")
    small = m.inverse(u, exact=True)
    d = np.abs(big - small)
    assert np.median(d) < 1e-6 and np.quantile(d, 0.999) < 2e-4 and d.max() < 2e-3
    u2, _ = m.flow(small)
    assert np.median(np.abs(u2 - u)) < 2e-5 and np.abs(u2 - u).max() < 5e-3


@pytest.mark.parametrize("D", [3, 5])
def test_mean_type_box_inverse_beyond_two_particles(D, monkeypatch):
    """The reference's reverse of the mean-type box transform is two-particle only (made.py:188); for D > 2 the library inverts
    direct_fun_mean itself: direct(inverse(u)) = u, walkers sorted and inside the box, in both sampler kernels."""
    from waveflow_amd import model_factory
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=5, i_spline_degree=5, n_prior_internal_knots=16,
                                                n_i_internal_knots=16, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6,
                                                n_flow_layers=2, box_size=7.0, xu_coord_type="mean")
    params, psi, log_pdf, sample = init_fun(3, D)
    m = psi.model
    m.ensure_params(params)
    u = np.random.default_rng(0).uniform(0.02, 0.98, size=(5000, D)).astype(np.float32)
    for limit in ("100000000", "0"):                                          # wave kernel, one-lane kernel
        monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", limit)
        x = m.inverse(u, exact=True)
        assert np.all(np.diff(x, axis=1) >= 0) and np.abs(x).max() <= 7.0 + 1e-4
        u2, _ = m.flow(x)
        assert np.median(np.abs(u2 - u)) < 3e-5 and np.abs(u2 - u).max() < 1e-2, (limit, np.abs(u2 - u).max())
    xs = m.sample(3, 4096, exact=True).cpu().numpy()
    assert np.all(np.diff(xs, axis=1) >= 0) and np.isfinite(psi(params, xs)).all()


def test_staged_sampler_of_large_batches(he_flat, monkeypatch):
    """From WF_SAMPLE_TILE_MIN walkers on (default 16 384) wf_sample / wf_inverse_fwd of the two-particle family run staged (wf_kernels_etile.hip:
    conditioners on the matrix cores, one lane per walker for the mesh searches): the inverse against the oracle and against the one-walker-per-wave
    kernel on the same latent points, the round trip through the forward pass, the draws against the wave kernel's (two-sample Kolmogorov-Smirnov),
    several passes over the model's scratch (a walker's stream is keyed by its index in the batch), and the fall-back of other models."""
    import torch
    from scipy import stats
    from waveflow_amd import model_factory
    params, psi, log_pdf, sample, om = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    g = np.random.default_rng(3)
    u = g.uniform(0.01, 0.99, size=(20001, 2)).astype(np.float32)
    for exact in (True, False):
        monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
        xs = m.inverse(u, exact=exact)
        # (round 4: the spline sums of the inverse read a window of 12 rows around the band of k + 1 that are neither 1 nor 0, + the prefix sum of the
        # coefficients left of it: the full rows give the same bits)
        monkeypatch.setenv("WF_SAMPLE_FULL_ROWS", "1")
        assert np.array_equal(xs, m.inverse(u, exact=exact))
        monkeypatch.delenv("WF_SAMPLE_FULL_ROWS")
        # ... and so do eight lanes per walker with eight-way mesh searches (k_tsample_p2g: the default with two row blocks, forced here)
        monkeypatch.setenv("WF_SAMPLE_GROUP_PHASE2", "1")
        assert np.array_equal(xs, m.inverse(u, exact=exact))
        monkeypatch.delenv("WF_SAMPLE_GROUP_PHASE2")
        monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")
        xw = m.inverse(u, exact=exact)                       # one wave per walker
        assert np.isfinite(xs).all() and not np.array_equal(xs, xw)
        d = np.abs(xs - xw)
        assert np.median(d) < 2e-6 and np.quantile(d, 0.999) < 4e-4 and d.max() < 2e-3, (exact, np.median(d), d.max())
        xo = om.inverse(he_flat, u[:3000], exact=exact)
        do = np.abs(xs[:3000] - xo)
        assert np.median(do) < 1e-5 and (do > 2e-3).mean() < 5e-3, (exact, np.median(do), do.max())
        if exact:
            u2, _ = m.flow(xs)
            assert np.median(np.abs(u2 - u)) < 2e-5 and np.abs(u2 - u).max() < 5e-3
    # the draws: reproducible, the same law as the wave kernel's, x = inverse(latent)
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
    xa, la = m.sample(11, 60000, return_latent=True, exact=True)
    assert torch.equal(xa, m.sample(11, 60000, exact=True)) and not torch.equal(xa, m.sample(12, 60000, exact=True))
    assert torch.isfinite(xa).all() and la.min().item() >= 0 and la.max().item() <= 1
    ub, _ = m.flow(xa)
    assert np.median(np.abs(ub.cpu().numpy() - la.cpu().numpy())) < 2e-5
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")
    xb, lb = m.sample(13, 60000, return_latent=True, exact=True)
    for c in range(2):
        for a, b in ((xa, xb), (la, lb)):
            p = stats.ks_2samp(a[:, c].cpu().numpy(), b[:, c].cpu().numpy()).pvalue
            assert p > 1e-4, (c, p)
    # the envelope of the prior's second column from the conditioner launch's o * keep (round 4: they ARE the plain B-spline coefficients where the boundary
    # map only zeroes coefficients) against the dense product e @ b_to_ob (WF_SAMPLE_DENSE_ENVELOPE): the same proposals, the same decisions but for
    # roundings of the bounds
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
    monkeypatch.setenv("WF_SAMPLE_DENSE_ENVELOPE", "1")
    xd = m.sample(11, 60000, exact=True)
    monkeypatch.delenv("WF_SAMPLE_DENSE_ENVELOPE")
    # (98 % are the same bits, 99.5 % the same proposal: the dense product forms each coefficient from 28 cancelling terms -- its bounds carry ~1e-5 of the
    # largest coefficient, enough to move one acceptance in a few thousand; the law is checked by the Kolmogorov-Smirnov lines above, which ran on the new form)
    close = ((xd - xa).abs().max(dim=1).values < 1e-4).float().mean().item()
    assert close > 0.99 and (xd == xa).all(dim=1).float().mean().item() > 0.9, close
    # ... and a proposal's value from the k + 1 plain B-splines alive on its knot interval (three table records per row) against the full rows of the
    # orthogonal table (WF_SAMPLE_FULL_ROWS): sum_i q_i b_i(x) = sum_j e_j ob_j(x) up to the fp32 roundings of the two tables
    monkeypatch.setenv("WF_SAMPLE_FULL_ROWS", "1")
    xf = m.sample(11, 60000, exact=True)
    monkeypatch.delenv("WF_SAMPLE_FULL_ROWS")
    closef = ((xf - xa).abs().max(dim=1).values < 1e-4).float().mean().item()
    assert closef > 0.995 and (xf == xa).all(dim=1).float().mean().item() > 0.99, closef
    # eight lanes per walker from the first proposal on (k_tsample_p1g) against the walker's own lane walking its proposals (WF_SAMPLE_ONE_LANE): the first
    # accepted proposal in sequence order either way -- the same draws, bit for bit
    monkeypatch.setenv("WF_SAMPLE_ONE_LANE", "1")
    xo, lo = m.sample(11, 60000, return_latent=True, exact=True)
    monkeypatch.delenv("WF_SAMPLE_ONE_LANE")
    assert torch.equal(xo, xa) and torch.equal(lo, la)
    # more walkers than one pass of the scratch holds (2^18): the same walkers as a prefix of the larger batch drew
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
    big = m.sample(21, (1 << 18) + 5000, exact=True)
    # (40 010 walkers end in a partial wave, whose lanes each finish their own rejection loop; in the larger batch the same walkers sit in a full
    # wave, where the unfinished ones are served by groups of eight lanes: the same draws either way)
    small = m.sample(21, 40010, exact=True)
    assert torch.equal(big[:40010], small) and torch.isfinite(big).all()
    tail = m.sample(21, (1 << 18) + 5000, exact=True)[(1 << 18):]
    assert torch.equal(tail, big[(1 << 18):]) and tail.std().item() > 0.5
    # a model outside the family (first-type box) keeps the other kernels: the switch changes nothing
    init = model_factory.get_waveflow_model(2, n_flow_layers=1, box_size=2, xu_coord_type="first")
    p1, psi1, lp1, _ = init(1, 2)
    lp1.model.ensure_params(p1)
    a = lp1.model.sample(5, 20000, exact=True)
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")
    assert torch.equal(a, lp1.model.sample(5, 20000, exact=True))


def test_staged_sampler_with_two_row_blocks(monkeypatch):
    """The staged large-batch sampler on 33 knots at k = 6 (39 / 38 bases: two 32-row blocks per dimension, BASELINE's "32-bin" variant): inverse
    against the one-walker-per-wave kernel on the same latent points, round trip through the forward pass, draws against the wave kernel's."""
    import torch
    from scipy import stats
    from waveflow_amd import model_factory
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=33,
                                                n_i_internal_knots=33, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6,
                                                n_flow_layers=2, box_size=10, xu_coord_type="mean")
    params, psi, log_pdf, sample = init_fun(3, 2)
    m = psi.model
    m.ensure_params(params)
    assert (m.i_nb, m.p_nb) == (39, 38)
    u = np.random.default_rng(5).uniform(0.01, 0.99, size=(20001, 2)).astype(np.float32)
    for exact in (True, False):
        monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
        xs = m.inverse(u, exact=exact)
        # (two row blocks: eight lanes per walker and eight-way mesh searches by default; the walker's own lane and the full table rows give the same bits)
        for switch in ("WF_SAMPLE_ONE_LANE", "WF_SAMPLE_FULL_ROWS"):
            monkeypatch.setenv(switch, "1")
            assert np.array_equal(xs, m.inverse(u, exact=exact)), switch
            monkeypatch.delenv(switch)
        monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")
        monkeypatch.setenv("WF_WAVE_SAMPLE_MAX", "100000000")
        xw = m.inverse(u, exact=exact)
        assert np.isfinite(xs).all() and not np.array_equal(xs, xw)
        d = np.abs(xs - xw)
        assert np.median(d) < 2e-6 and np.quantile(d, 0.999) < 4e-4 and d.max() < 2e-3, (exact, np.median(d), d.max())
        if exact:
            u2, _ = m.flow(xs)
            assert np.median(np.abs(u2 - u)) < 2e-5 and np.abs(u2 - u).max() < 5e-3
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
    xa, la = m.sample(11, 60001, return_latent=True, exact=True)
    assert torch.equal(xa, m.sample(11, 60001, exact=True)) and torch.isfinite(xa).all() and torch.isfinite(la).all()
    monkeypatch.setenv("WF_SAMPLE_ONE_LANE", "1")      # (the same draws from the walker's own lane)
    assert torch.equal(xa, m.sample(11, 60001, exact=True))
    monkeypatch.delenv("WF_SAMPLE_ONE_LANE")
    ub, _ = m.flow(xa)
    assert np.median(np.abs(ub.cpu().numpy() - la.cpu().numpy())) < 2e-5
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")
    xb, lb = m.sample(13, 60000, return_latent=True, exact=True)
    for c in range(2):
        for a, b in ((xa, xb), (la, lb)):
            p = stats.ks_2samp(a[:, c].cpu().numpy(), b[:, c].cpu().numpy()).pvalue
            assert p > 1e-4, (c, p)


def test_staged_sampler_other_models(monkeypatch):
    """The staged sampler on models with derivative boundary constraints on the layers and the prior, one and two layers, other boxes, degrees and
    knot counts: inverse against the one-walker-per-wave kernel, round trip, draws against the wave kernel's."""
    import torch
    from scipy import stats
    from waveflow_amd import flows, model_factory, wavefunctions
    mt = model_factory.get_masked_transform
    il, ir, pl, pr = {0: 0.0, 1: 0.0}, {0: 1.0, 1: 0.0}, {0: 0, 2: 0}, {0: 0, 1: 0}
    cases = [
        dict(L=3.0, n=2, k=6, kn=23, il=il, ir=ir, pl=pl, pr=pr),
        dict(L=2.0, n=1, k=5, kn=16, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}),
        dict(L=6.0, n=3, k=3, kn=10, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}),
    ]
    g = np.random.default_rng(9)
    for c in cases:
        init = wavefunctions.Waveflow(
            flows.Serial(flows.BoxTransformLayer(c["L"]), *(flows.IMADE(mt(), c["k"], c["kn"], 0.05, 1e-6, c["il"], c["ir"]), flows.Reverse()) * c["n"]),
            mt(allow_negative_params=True), c["k"], c["kn"], constraints_dict_left=c["pl"], constraints_dict_right=c["pr"],
            constrained_dimension_indices_left=[0], set_nn_output_grad_to_zero=False)
        params, psi, log_pdf, _ = init(4, 2)
        m = psi.model
        m.ensure_params(params)
        u = g.uniform(0.01, 0.99, size=(17000, 2)).astype(np.float32)
        monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
        xs = m.inverse(u, exact=True)
        xa, la = m.sample(3, 30000, return_latent=True, exact=True)
        monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")
        xw = m.inverse(u, exact=True)
        xb, lb = m.sample(4, 30000, return_latent=True, exact=True)
        assert np.isfinite(xs).all() and not np.array_equal(xs, xw), c
        d = np.abs(xs - xw) / c["L"]
        assert np.median(d) < 1e-6 and np.quantile(d, 0.999) < 1e-4 and d.max() < 1e-3, (c, np.median(d), d.max())
        u2, _ = m.flow(xs)
        assert np.median(np.abs(u2 - u)) < 2e-5, c
        assert torch.isfinite(xa).all() and torch.isfinite(la).all(), c
        for col in range(2):
            for a, b in ((xa, xb), (la, lb)):
                p = stats.ks_2samp(a[:, col].cpu().numpy(), b[:, col].cpu().numpy()).pvalue
                assert p > 1e-4, (c, col, p)


def test_staged_sampler_leaves_high_prior_degrees_to_the_wave_sampler(monkeypatch):
    """k_tsample's piecewise-constant envelope of the second prior column takes the maximum over at most 9 coefficients per
    knot interval; a prior of degree > 8 has more live B-splines per interval, so those models keep the wave sampler (whose bound is the
    global one) at every batch size: same seed, same draws with the staged path switched on and off -- and the draws follow the oracle's
    column density."""
    import torch
    from scipy import stats
    from waveflow_amd import flatten_params, flows, model_factory, wavefunctions
    mt = model_factory.get_masked_transform
    init = wavefunctions.Waveflow(
        flows.Serial(flows.BoxTransformLayer(3.0), flows.IMADE(mt(), 5, 16, 0.05, 1e-6, {0: 0.0}, {0: 1.0}), flows.Reverse()),
        mt(allow_negative_params=True), 9, 14, constraints_dict_left={0: 0}, constraints_dict_right={0: 0},
        constrained_dimension_indices_left=[0], set_nn_output_grad_to_zero=False)
    params, psi, log_pdf, _ = init(6, 2)
    m = psi.model
    m.ensure_params(params)
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "16384")
    xa, la = m.sample(3, 20000, return_latent=True, exact=True)
    monkeypatch.setenv("WF_SAMPLE_TILE_MIN", "0")
    xb, lb = m.sample(3, 20000, return_latent=True, exact=True)
    assert torch.isfinite(xa).all() and torch.equal(xa, xb) and torch.equal(la, lb)
    # first latent column against the oracle's density of that column (bsplines_jax.py:144-171)
    om = oracle.Model(D=2, n_layers=1, box="mean", box_L=3.0, i_k=5, i_knots=16, i_reg=0.05, prior="waveflow", p_k=9, p_knots=14, constr_left=(0,))
    flat = flatten_params(params)
    grid = np.linspace(0.0, 1.0, 4001).astype(np.float32)
    dens, _ = om.prior_column_density(flat, np.zeros(2, np.float32), 0, grid)
    cdf = np.cumsum(dens.astype(np.float64)); cdf /= cdf[-1]
    p = stats.kstest(la[:, 0].cpu().numpy().astype(np.float64), lambda t: np.interp(t, grid, cdf)).pvalue
    assert p > 1e-4, p
