"""Local energy H psi = -1/2 laplacian(psi) + V psi on the HIP path vs oracle/energy_torch.py."""
import numpy as np
import pytest

import oracle
from conftest import sorted_walkers

pytestmark = pytest.mark.gpu


def he(he_flat):
    from waveflow_amd import checkpoint, model_factory
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10)
    params, psi, log_pdf, sample = init_fun(0, 2)
    return checkpoint.unflatten_like(params, he_flat), psi, log_pdf, sample


def test_hamiltonian_vs_autograd_oracle(golden, he_flat):
    import torch
    from oracle import energy_torch as et
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    protons, n_el = physics.system_catalogue[1]["He"]
    assert n_el == 2
    h_fn = physics.construct_hamiltonian_function(psi, protons=protons, n_space_dimensions=1, eps=0.0)
    x = np.concatenate([np.sort(golden["he_golden"]["sample_points"], -1), sorted_walkers(250, 2, 10.0, 5)]).astype(np.float32)
    h = h_fn(params, x)
    assert h.shape == (500, 1)
    hp, ps, lap = h_fn.model.hamiltonian(x, protons.reshape(-1), return_psi=True, return_laplacian=True)
    ho64, po64, lo64 = et.hamiltonian(et.he_model(torch.float64), he_flat, x.astype(np.float64), protons.reshape(-1))
    ho32, po32, lo32 = et.hamiltonian(et.he_model(torch.float32), he_flat, x, protons.reshape(-1))
    np.testing.assert_allclose(ps, po64, rtol=0, atol=3e-5)
    # second derivatives amplify fp32 rounding: compare with the fp64 autograd oracle, using the fp32 autograd
    # oracle's own deviation as the yardstick (same idea as tests/test_gpu_parity.py)
    scale = np.abs(lo64).max()
    e_g, e_o = np.abs(lap - lo64), np.abs(lo32 - lo64)
    # (the wave kernel on this 500-walker batch: measured 2.2 x the fp32 torch oracle's own worst deviation from fp64, median 0.9 x)
    assert np.median(e_g) <= 3 * np.median(e_o) + 1e-6 * scale, (np.median(e_g), np.median(e_o))
    assert e_g.max() <= 4 * e_o.max() + 2e-5 * scale, (e_g.max(), e_o.max(), scale)
    np.testing.assert_allclose(hp, ho64, rtol=0, atol=4 * np.abs(ho32 - ho64).max() + 2e-5 * np.abs(ho64).max())
    # local energy as vqmc.loss_fn_efficient forms it (vqmc.py:193-200)
    el = hp / (ps + 1e-8)
    assert np.isfinite(el).all()


def test_hamiltonian_errors_and_shapes(he_flat):
    from waveflow_amd import _lib, model_factory
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    assert m.hamiltonian(np.zeros((0, 2), np.float32), [0.0]).shape == (0,)
    L = _lib.lib()
    assert L.wf_hamiltonian_fwd(m._h, None, 4, None, 9, None, None, None, None) == -1
    p2, lp2, _ = model_factory.get_model(n_flow_layers=1)(0, 2)
    lp2.model.ensure_params(p2)
    x = np.random.default_rng(0).uniform(0.1, 0.9, size=(8, 2)).astype(np.float32)
    with pytest.raises(_lib.WfError):
        lp2.model.hamiltonian(x, [0.0])


def test_checkpoint_artefacts_match_the_reference_files(tmp_path, golden, he_flat):
    """helpers.create_checkpoint_wavefunc on the HIP path reproduces the files the reference wrote for the same checkpoint."""
    import pickle
    from waveflow_amd import checkpoint, flatten_params
    from waveflow_amd.utils import helpers
    params, psi, log_pdf, sample = he(he_flat)
    system_dict = {"system_name": "He", "box_length": 10, "n_particle": 2, "n_space_dimension": 1, "window": 100, "n_plotting": 200}
    helpers.create_checkpoint_wavefunc(3, str(tmp_path), psi, sample, params, 100000, [0.0, -1.5], [[-1.5]], system_dict)
    g = golden["he_golden"]
    z = np.load(tmp_path / "outputs/wavefunctions_2d/values_epoch100000.npy")
    assert z.shape == g["psi_grid"].shape and z.dtype == g["psi_grid"].dtype
    assert np.abs(z - g["psi_grid"]).max() < 2.5e-5
    on = np.load(tmp_path / "outputs/density_1e/onproton_values_epoch100000.npy")
    assert np.abs(on - g["onproton_values"]).max() < 1e-5
    np.testing.assert_allclose(np.load(tmp_path / "outputs/density_1e/onproton_coord_epoch100000.npy"), g["onproton_coord"], atol=1e-6)
    sp = np.load(tmp_path / "outputs/sample_points/values_epoch100000.npy")
    assert sp.shape == g["sample_points"].shape and sp.dtype == g["sample_points"].dtype
    assert np.load(tmp_path / "loss.npy").shape == (2,)
    # the checkpoint round-trips through the reference-format loader
    loaded, epoch = checkpoint.load_reference_checkpoint(str(tmp_path / "checkpoints"))
    assert epoch == 100000 and np.array_equal(flatten_params(loaded), he_flat)
    with open(tmp_path / "checkpoints", "rb") as f:
        plain, _ = pickle.load(f)       # plain pickle.load works too: leaves are NumPy arrays
    assert np.array_equal(flatten_params(plain), he_flat)


@pytest.mark.parametrize("D", [2, 3, 5, 8])
def test_forward_laplacian_sweep_matches_directional_sweeps(D, monkeypatch):
    """wf_hamiltonian_fwd carries (value, gradient, Laplacian) per walker in one pass; WF_ENERGY_R3 switches back to D passes
    of second-order Taylor coefficients (the sweep the gradient path tapes).  Same psi bit for bit, same Laplacian up to
    rounding."""
    from waveflow_amd import model_factory
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=2, box_size=10.0)
    params, psi, log_pdf, sample = init_fun(11, D)
    m = psi.model
    m.ensure_params(params)
    x = sorted_walkers(777, D, 10.0, 3).astype(np.float32)
    protons = np.linspace(-3, 3, D)
    monkeypatch.setenv("WF_ENERGY_R3", "1")
    h3, p3, l3 = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
    monkeypatch.delenv("WF_ENERGY_R3")
    hf, pf, lf = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
    assert np.array_equal(p3, pf)
    assert np.isfinite(lf).all() and np.abs(lf).max() > 0
    assert np.linalg.norm(lf - l3) <= 1e-4 * np.linalg.norm(l3), np.linalg.norm(lf - l3) / np.linalg.norm(l3)
    assert np.linalg.norm(hf - h3) <= 1e-4 * np.linalg.norm(h3)


def test_laplacian_where_the_prior_output_sum_vanishes():
    """A walker met in a batch-4096 He run (scratch/nan_hunt3.py wrote the fixture: parameters of step 111, one walker): the raw
    outputs of the prior conditioner for dimension 1 sum to 2.6e-7.  The reference divides by that sum before it L2-normalises
    (model_factory.py:69, bsplines_jax.py:130); in fp32 the quotient form gives a Laplacian of -5e2 (torch oracle) to -7e3 or NaN,
    the true value (fp64 oracle) is 0.088.  The kernels evaluate the equivalent sign form and stay at the fp64 value."""
    import os
    import torch
    from conftest import GOLDEN
    from oracle import energy_torch as et
    from waveflow_amd import checkpoint, model_factory
    z = np.load(os.path.join(GOLDEN, "he_prior_sum_zero.npz"))
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10)
    params, psi, log_pdf, sample = init_fun(0, 2)
    params = checkpoint.unflatten_like(params, z["flat"])
    m = psi.model
    m.ensure_params(params)
    x = np.repeat(z["x"], 64, 0)
    protons = np.zeros(2)
    h64, p64, l64 = et.hamiltonian(et.he_model(torch.float64), z["flat"], z["x"].astype(np.float64), protons)
    assert abs(l64[0]) < 1.0
    h, ps, lap = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
    # (the sign of psi is the sign of that sum -- noise at this point, and the local energy does not depend on it)
    assert np.isfinite(h).all() and np.abs(lap / ps - l64[0] / p64[0]).max() < 5e-2 * abs(l64[0] / p64[0]), (lap[0], ps[0], l64[0], p64[0])
    np.testing.assert_allclose(np.abs(ps), abs(p64[0]), rtol=1e-4)
    np.testing.assert_allclose(h / ps, h64[0] / p64[0], rtol=5e-2)
    sums, grad = m.vqmc_loss_grad(x, protons, -1.0)
    assert torch.isfinite(grad).all() and np.isfinite(sums.cpu().numpy()).all()
    el = h / (ps + 1e-8)
    np.testing.assert_allclose(sums.cpu().numpy()[0] / 64, el.astype(np.float64).mean(), rtol=1e-3)


def _tile_and_wave(model, x, protons, monkeypatch):
    """(H psi, psi, laplacian) of x through the matrix-core tile path (forced) and through the wave kernel (tile path disabled)."""
    out = []
    for tile_min in ("1", "0"):
        monkeypatch.setenv("WF_ENERGY_TILE_MIN", tile_min)
        out.append([np.asarray(t, dtype=np.float64) for t in model.hamiltonian(x, protons, return_psi=True, return_laplacian=True)])
    monkeypatch.delenv("WF_ENERGY_TILE_MIN")
    return out


def test_local_energy_on_the_matrix_cores_vs_oracle_and_wave_kernel(golden, he_flat, monkeypatch):
    """wf_hamiltonian_fwd of large two-particle batches: conditioner Taylor channels on the matrix cores + lane-per-walker heads
    (wf_kernels_etile.hip).  Same yardstick as test_hamiltonian_vs_autograd_oracle for the oracle; against the wave kernel on a large,
    ragged batch; the switch itself (default threshold, WF_ENERGY_TILE_MIN)."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    protons = physics.system_catalogue[1]["He"][0].reshape(-1)
    x = np.concatenate([np.sort(golden["he_golden"]["sample_points"], -1), sorted_walkers(250, 2, 10.0, 5)]).astype(np.float32)
    (hp, ps, lap), (hw, pw, lw) = _tile_and_wave(m, x, protons, monkeypatch)
    ho64, po64, lo64 = et.hamiltonian(et.he_model(torch.float64), he_flat, x.astype(np.float64), protons)
    ho32, po32, lo32 = et.hamiltonian(et.he_model(torch.float32), he_flat, x, protons)
    np.testing.assert_allclose(ps, po64, rtol=0, atol=3e-5)
    scale = np.abs(lo64).max()
    e_g, e_o = np.abs(lap - lo64), np.abs(lo32 - lo64)
    # (round 3: the one-kernel form k_efused; measured 1.6 x the fp32 torch oracle's own worst deviation from fp64, median 1.6 x)
    assert np.median(e_g) <= 3 * np.median(e_o) + 1e-6 * scale, (np.median(e_g), np.median(e_o))
    assert e_g.max() <= 3 * e_o.max() + 2e-5 * scale, (e_g.max(), e_o.max(), scale)
    np.testing.assert_allclose(hp, ho64, rtol=0, atol=3 * np.abs(ho32 - ho64).max() + 2e-5 * np.abs(ho64).max())
    # the launch-per-net form of the same path (conditioner kernel -> exchange buffer -> head kernel; what models whose nets do not fit LDS
    # together take): same oracle bound, and the two forms agree
    monkeypatch.setenv("WF_ENERGY_FUSED", "0")
    (hq, pq, lq), _ = _tile_and_wave(m, x, protons, monkeypatch)
    monkeypatch.delenv("WF_ENERGY_FUSED")
    assert np.abs(lq - lo64).max() <= 3 * e_o.max() + 2e-5 * scale and not np.array_equal(lq, lap)
    for a, b in ((pq, ps), (lq, lap), (hq, hp)):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()
    # a large batch that is not a multiple of the 32-walker tile, walkers up to the box edge (clipped prior arguments included)
    xb = sorted_walkers(70001, 2, 10.0, 21)
    (hp, ps, lap), (hw, pw, lw) = _tile_and_wave(m, xb, protons, monkeypatch)
    assert np.isfinite(hp).all()
    for a, b in ((ps, pw), (lap, lw), (hp, hw)):
        d = np.abs(a - b)
        assert d.max() <= 2e-4 * np.abs(b).max() and np.median(d) <= 1e-7 * np.abs(b).max(), (d.max() / np.abs(b).max(), np.median(d) / np.abs(b).max())
    # default switch: 70 001 walkers take the tile path (same numbers as forcing it), 5 000 the wave kernel
    h_default = np.asarray(m.hamiltonian(xb, protons), dtype=np.float64)
    assert np.array_equal(h_default, hp)
    # launch after launch the same bits (the conditioner kernel mixes MFMA chains with VALU work like k_mfma: same build rule)
    import torch
    xt = torch.as_tensor(sorted_walkers(1 << 18, 2, 10.0, 77)).cuda()
    first = m.hamiltonian(xt, protons).clone()
    for _ in range(12):
        assert torch.equal(m.hamiltonian(xt, protons), first)
    h_small = np.asarray(m.hamiltonian(xb[:5000], protons), dtype=np.float64)
    assert np.array_equal(h_small, hw[:5000])


def test_local_energy_tile_path_other_models(monkeypatch):
    """The tile path on models the He checkpoint does not exercise: derivative boundary constraints (the regrouped tables carry the
    boundary map), one and two layers, another box and knot count; a model outside the family (first-type box) keeps the wave kernel."""
    from waveflow_amd import flows, model_factory, wavefunctions
    mt = model_factory.get_masked_transform
    il, ir, pl, pr = {0: 0.0, 1: 0.0}, {0: 1.0, 1: 0.0}, {0: 0, 2: 0}, {0: 0, 1: 0}
    cases = [
        dict(L=3.0, n=2, k=6, kn=23, il=il, ir=ir, pl=pl, pr=pr),
        dict(L=2.0, n=1, k=5, kn=16, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}),
        # 33 knots = 39 / 38 bases: two 32-row blocks per dimension (BASELINE's "32-bin" variant of C3) -- the one-kernel form only
        dict(L=10.0, n=3, k=6, kn=33, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}),
    ]
    for c in cases:
        init = wavefunctions.Waveflow(
            flows.Serial(flows.BoxTransformLayer(c["L"]), *(flows.IMADE(mt(), c["k"], c["kn"], 0.05, 1e-6, c["il"], c["ir"]), flows.Reverse()) * c["n"]),
            mt(allow_negative_params=True), c["k"], c["kn"], constraints_dict_left=c["pl"], constraints_dict_right=c["pr"],
            constrained_dimension_indices_left=[0], set_nn_output_grad_to_zero=False)
        params, psi, log_pdf, _ = init(4, 2)
        m = psi.model
        m.ensure_params(params)
        x = sorted_walkers(4097, 2, 0.95 * c["L"], 13)
        (hp, ps, lap), (hw, pw, lw) = _tile_and_wave(m, x, [0.0, 0.0], monkeypatch)
        for a, b in ((ps, pw), (lap, lw), (hp, hw)):
            d = np.abs(a - b)
            assert np.isfinite(a).all() and d.max() <= 5e-4 * np.abs(b).max() and np.median(d) <= 1e-6 * np.abs(b).max(), (c, d.max() / np.abs(b).max())
        # the head kernels read every table chunk at the mesh index clamped to the chunk's support (wf_model.cpp: upload_chunked): the same
        # bits as the reads at the walker's own index (a model created with the clamp switched off)
        monkeypatch.setenv("WF_MFMA_NO_BAND", "1")
        params2, psi2, _, _ = init(4, 2)
        psi2.model.ensure_params(params2)
        monkeypatch.delenv("WF_MFMA_NO_BAND")
        (hp2, ps2, lap2), _ = _tile_and_wave(psi2.model, x, [0.0, 0.0], monkeypatch)
        assert np.array_equal(hp2, hp) and np.array_equal(ps2, ps) and np.array_equal(lap2, lap)
    # first-type box: not in the family -- forcing the tile path changes nothing
    init = model_factory.get_waveflow_model(2, n_flow_layers=1, box_size=2, xu_coord_type="first")
    params, psi, log_pdf, _ = init(1, 2)
    m = psi.model
    m.ensure_params(params)
    x = sorted_walkers(512, 2, 1.9, 3)
    (hp, _, _), (hw, _, _) = _tile_and_wave(m, x, [0.0, 0.0], monkeypatch)
    assert np.array_equal(hp, hw)


@pytest.mark.parametrize("D", [3, 4, 8])
def test_local_energy_on_the_matrix_cores_beyond_two_particles(D, monkeypatch):
    """wf_hamiltonian_fwd of large batches of D-particle models: one coordinate direction at a time, Taylor triples on the matrix
    cores (wf_kernels_etile_dir.hip).  Against the fp64 torch oracle with the yardstick of the two-particle test (3 x the fp32 torch oracle's own
    deviation); against the wave kernel on a ragged batch up to the box edge; the switch; launch-to-launch bits."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, model_factory
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=2 if D < 8 else 3, box_size=10.0)
    params, psi, log_pdf, sample = init_fun(11, D)
    m = psi.model
    m.ensure_params(params)
    flat = flatten_params(params)
    protons = np.linspace(-3, 3, D)
    n_layers = 2 if D < 8 else 3
    x = sorted_walkers(300 if D < 8 else 120, D, 9.5, 3).astype(np.float32)
    (hp, ps, lap), (hw, pw, lw) = _tile_and_wave(m, x, protons, monkeypatch)
    assert not np.array_equal(lap, lw)                                  # the forced path is another kernel
    mk = lambda dt: et.TorchWaveflow(D, n_layers, "mean", 10.0, 6, 23, 0.05, tuple(range(D - 1)), dtype=dt)
    ho64, po64, lo64 = et.hamiltonian(mk(torch.float64), flat, x.astype(np.float64), protons)
    ho32, po32, lo32 = et.hamiltonian(mk(torch.float32), flat, x, protons)
    np.testing.assert_allclose(ps, po64, rtol=0, atol=3e-5 * np.abs(po64).max() + 3 * np.abs(po32 - po64).max())
    scale = np.abs(lo64).max()
    e_g, e_o = np.abs(lap - lo64), np.abs(lo32 - lo64)
    print(f"[hpsi D={D}] laplacian vs fp64 torch oracle: tile max {e_g.max():.2e} median {np.median(e_g):.2e}; fp32 torch oracle max {e_o.max():.2e} median "
          f"{np.median(e_o):.2e}; wave kernel max {np.abs(lw - lo64).max():.2e}; scale {scale:.2e}")
    assert np.median(e_g) <= 3 * np.median(e_o) + 1e-6 * scale, (np.median(e_g), np.median(e_o))
    assert e_g.max() <= 3 * e_o.max() + 2e-5 * scale, (e_g.max(), e_o.max(), scale)
    np.testing.assert_allclose(hp, ho64, rtol=0, atol=3 * np.abs(ho32 - ho64).max() + 2e-5 * np.abs(ho64).max())
    # a ragged batch, walkers up to the box edge (clipped prior arguments included): against the wave kernel
    xb = sorted_walkers(20001, D, 10.0, 21)
    (hp, ps, lap), (hw, pw, lw) = _tile_and_wave(m, xb, protons, monkeypatch)
    assert np.isfinite(hp).all()
    for a, b in ((ps, pw), (lap, lw), (hp, hw)):
        d = np.abs(a - b)
        assert d.max() <= 5e-4 * np.abs(b).max() and np.median(d) <= 1e-6 * np.abs(b).max(), (D, d.max() / np.abs(b).max(), np.median(d) / np.abs(b).max())
    # default switch: 20 001 walkers take the tile path (same numbers as forcing it), 5 000 the wave kernel
    assert np.array_equal(np.asarray(m.hamiltonian(xb, protons), dtype=np.float64), hp)
    assert np.array_equal(np.asarray(m.hamiltonian(xb[:5000], protons), dtype=np.float64), hw[:5000])
    xt = torch.as_tensor(xb).cuda()
    first = m.hamiltonian(xt, protons).clone()
    for _ in range(4):
        assert torch.equal(m.hamiltonian(xt, protons), first)
