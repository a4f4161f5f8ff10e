"""Parameter gradients on the HIP path (vqmc.train_step_efficient) vs oracle/energy_torch.py (torch reverse
mode through the Hessian trace, fp64)."""
import numpy as np
import pytest

from conftest import sorted_walkers

pytestmark = pytest.mark.gpu


def he(he_flat):
    from waveflow_amd import checkpoint, model_factory
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10)
    params, psi, log_pdf, sample = init_fun(0, 2)
    return checkpoint.unflatten_like(params, he_flat), psi, log_pdf, sample


def rel_l2(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


# Bounds of the gradient comparisons against the fp64 torch oracle (reverse mode through the Hessian trace): 3 x the value measured on an MI355X in
# round 4 (profiles/r04_grad_rel_l2_measured.txt; rounds 1 - 3 had 5e-3 everywhere, 40 - 300 x the measured values: a wrong small leaf could hide).
# The key is the line of the assertion in round 3's file, kept as a stable name.
# measured: L59 1.19e-4 (loss_fn_efficient's gradient: the seeds carry E_L - <E_L>), L82 1.53e-5, L87 1.72e-5, L314 2.02e-5, L803 2.8e-6, L871 3.9e-6, L963 <= 4.9e-6
BOUND = {"L59": 4e-4, "L82": 5e-5, "L87": 6e-5, "L314": 7e-5, "L803": 1e-5, "L871": 1.2e-5, "L963": 1.5e-5}


def rel_ok(got, want, bound, what):
    r = rel_l2(got, want)
    print(f"[rel_l2 {what}] measured {r:.3e} (bound {bound:g})")
    assert r < bound, (what, r, bound)
    return r


def test_psi_vjp_vs_autograd_oracle(golden, he_flat):
    import torch
    from oracle import energy_torch as et
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    x = np.concatenate([np.sort(golden["he_golden"]["sample_points"], -1)[:64], sorted_walkers(64, 2, 8.0, 5)]).astype(np.float32)
    g = np.random.default_rng(3)
    mo = et.he_model(torch.float64)
    # psi only, Laplacian only, both
    for w_psi, w_lap in ((g.normal(size=128), np.zeros(128)), (np.zeros(128), g.normal(size=128)), (g.normal(size=128), g.normal(size=128))):
        got = m.psi_vjp(x, w_psi.astype(np.float32), w_lap.astype(np.float32)).cpu().numpy().astype(np.float64)
        want = et.psi_vjp(mo, he_flat, x.astype(np.float64), w_psi.astype(np.float32), w_lap.astype(np.float32))
        assert got.shape == want.shape == (he_flat.size,)
        assert np.isfinite(got).all()
        # entries the masks remove and the zero_params leaves carry no gradient on either side
        assert ((want == 0) <= (np.abs(got) <= 1e-12)).all()
        assert rel_l2(got, want) < 2e-3, rel_l2(got, want)
        big = np.abs(want) > 1e-3 * np.abs(want).max()
        assert np.abs(got[big] / want[big] - 1).max() < 5e-2


def test_vqmc_loss_grad_vs_oracle(golden, he_flat):
    import torch
    from oracle import energy_torch as et
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    protons, _ = physics.system_catalogue[1]["He"]
    x = np.sort(golden["he_golden"]["sample_points"], -1)[:192].astype(np.float32)
    sums, grad = m.vqmc_loss_grad(x, protons.reshape(-1), running_average=-2.5)
    sums = sums.cpu().numpy()
    loss = sums[0] / sums[2]
    lo, go, elo = et.vqmc_loss_grad(et.he_model(torch.float64), he_flat, x.astype(np.float64), protons.reshape(-1), -2.5)
    assert abs(loss - lo) < 1e-3 * max(1.0, np.abs(elo).mean())
    rel_ok(grad.cpu().numpy().astype(np.float64), go, BOUND["L59"], "L59")


def test_gradients_on_the_matrix_cores_vs_oracle_and_wave_sweeps(golden, he_flat, monkeypatch):
    """value_and_grad(loss_fn_efficient) (vqmc.py:193-221) for large two-particle batches: forward, per-net reverse kernels and weight-gradient
    products on the matrix cores (wf_kernels_etile.hip: k_efused, k_ebwd; head pullbacks: wf_etile_adjoint.h, checked on the CPU by
    tests/test_etile_adjoint.py).  Forced on a small batch (partial tile included) against the torch autograd oracle; on 50 001 walkers against the
    wave sweeps (psi_vjp with random weights, and the loss + gradient entry point on walkers from the model's own sampler); several chunks; the
    switch itself; bitwise reproducibility."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import _lib
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    protons = physics.system_catalogue[1]["He"][0].reshape(-1)
    x = np.sort(golden["he_golden"]["sample_points"], -1)[:171].astype(np.float32)       # 5 tiles + 11 walkers
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
    sums, grad = m.vqmc_loss_grad(x, protons, running_average=-2.5)
    lo, go, elo = et.vqmc_loss_grad(et.he_model(torch.float64), he_flat, x.astype(np.float64), protons, -2.5)
    s = sums.cpu().numpy()
    assert abs(s[0] / s[2] - lo) < 1e-3 * max(1.0, np.abs(elo).mean())
    rel_ok(grad.cpu().numpy().astype(np.float64), go, BOUND["L82"], "L82")
    g = np.random.default_rng(5)
    w1, w2 = g.normal(size=len(x)).astype(np.float32), (0.1 * g.normal(size=len(x))).astype(np.float32)
    got = m.psi_vjp(x, w1, w2).cpu().numpy().astype(np.float64)
    want = et.psi_vjp(et.he_model(torch.float64), he_flat, x.astype(np.float64), w1, w2)
    rel_ok(got, want, BOUND["L87"], "L87")
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "0")
    got_w = m.psi_vjp(x, w1, w2).cpu().numpy().astype(np.float64)
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
    assert not np.array_equal(got, got_w) and rel_l2(got, got_w) < 5e-4      # (the forced small batch did take the matrix-core path: another kernel's bits)
    # a large ragged batch: against the wave sweeps (fp32 both: 1.6e-5 measured), bit-reproducible, and in several chunks
    xb = torch.as_tensor(sorted_walkers(50001, 2, 9.5, 7)).cuda()
    wb1 = torch.as_tensor(g.normal(size=50001).astype(np.float32)).cuda()
    wb2 = torch.as_tensor((0.1 * g.normal(size=50001)).astype(np.float32)).cuda()
    tile = m.psi_vjp(xb, wb1, wb2)
    assert torch.equal(tile, m.psi_vjp(xb, wb1, wb2))
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "0")
    wave = m.psi_vjp(xb, wb1, wb2)
    monkeypatch.delenv("WF_GRAD_TILE_MIN")
    assert torch.equal(m.psi_vjp(xb, wb1, wb2), tile)                      # default switch: 50 001 walkers take the matrix-core path
    assert rel_l2(tile.cpu().numpy(), wave.cpu().numpy()) < 2e-4 and not torch.equal(tile, wave)
    L = _lib.lib()
    ws = torch.empty(40 * 1024 * 1024, device="cuda", dtype=torch.uint8)    # room for ~ 9 000 walkers per chunk
    grad = torch.empty(m.n_params, device="cuda")
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
    rc = L.wf_psi_vjp(m._h, xb.data_ptr(), 50001, wb1.data_ptr(), wb2.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_l2(grad.cpu().numpy(), tile.cpu().numpy()) < 1e-4
    # the loss + gradient entry point on walkers from |psi|^2 (where E_L is well conditioned)
    xs = m.sample(11, 40000, exact=True)
    st, gt = m.vqmc_loss_grad(xs, protons, -1.8)
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "0")
    sw, gw = m.vqmc_loss_grad(xs, protons, -1.8)
    monkeypatch.delenv("WF_GRAD_TILE_MIN")
    np.testing.assert_allclose(st.cpu().numpy(), sw.cpu().numpy(), rtol=1e-4)
    assert rel_l2(gt.cpu().numpy(), gw.cpu().numpy()) < 2e-3, rel_l2(gt.cpu().numpy(), gw.cpu().numpy())


@pytest.mark.gpu
def test_gradients_on_the_matrix_cores_with_two_row_blocks_vs_autograd_oracle(monkeypatch):
    """The 33-knot ("32-bin") variant of C3 -- 39 / 38 bases per dimension, two 32-row blocks of every head -- through the matrix-core gradient
    path (k_ebwd<., 2>; round 3 left it to the wave sweeps: 8.1 ms per 2^17 walkers): psi / Laplacian weights and value_and_grad(loss_fn_efficient)
    (vqmc.py:193-221) forced on a small ragged batch against the torch autograd oracle (fp64), a large ragged batch against the wave sweeps, the
    default switch, bitwise reproducibility (the workgroup's shared accumulators take the tiles in a fixed order)."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, model_factory
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=33, n_i_internal_knots=33,
                                                i_spline_reg=0.05, n_flow_layers=3, box_size=10.0)
    params, psi, log_pdf, sample = init_fun(3, 2)
    m = psi.model
    m.ensure_params(params)
    flat = flatten_params(params)
    mo = et.TorchWaveflow(2, 3, "mean", 10.0, 6, 33, 0.05, (0,), dtype=torch.float64)
    x = sorted_walkers(171, 2, 9.0, 21)
    g = np.random.default_rng(8)
    wp, wl = g.normal(size=171).astype(np.float32), (0.1 * g.normal(size=171)).astype(np.float32)
    want = et.psi_vjp(mo, flat, x.astype(np.float64), wp, wl)
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
    got = m.psi_vjp(x, wp, wl).cpu().numpy().astype(np.float64)
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "0")
    got_w = m.psi_vjp(x, wp, wl).cpu().numpy().astype(np.float64)
    assert not np.array_equal(got, got_w)                      # the forced small batch did take the matrix-core path
    # against the fp64 oracle the seeded 33-knot model sits at 2.4e-3 on EITHER fp32 path (the bound of test_gradients_other_shapes_vs_autograd_oracle);
    # the two paths agree leaf by leaf to 2.4e-6 (scratch/r04_grad33_diag.py)
    assert rel_l2(got, want) < 3e-3, rel_l2(got, want)
    assert rel_l2(got_w, want) < 3e-3, rel_l2(got_w, want)
    assert rel_l2(got, got_w) < 5e-5, rel_l2(got, got_w)
    protons = np.array([0.0, 0.0])
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
    sums, grad = m.vqmc_loss_grad(x, protons, running_average=-2.5)
    lo, go, elo = et.vqmc_loss_grad(mo, flat, x.astype(np.float64), protons, -2.5)
    s = sums.cpu().numpy()
    assert abs(s[0] / s[2] - lo) < 1e-3 * max(1.0, np.abs(elo).mean())
    assert rel_l2(grad.cpu().numpy().astype(np.float64), go) < 3e-3, rel_l2(grad.cpu().numpy().astype(np.float64), go)
    xb = torch.as_tensor(sorted_walkers(50001, 2, 9.5, 7)).cuda()
    wb1 = torch.as_tensor(g.normal(size=50001).astype(np.float32)).cuda()
    wb2 = torch.as_tensor((0.1 * g.normal(size=50001)).astype(np.float32)).cuda()
    tile = m.psi_vjp(xb, wb1, wb2)
    assert torch.equal(tile, m.psi_vjp(xb, wb1, wb2))
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "0")
    wave = m.psi_vjp(xb, wb1, wb2)
    monkeypatch.delenv("WF_GRAD_TILE_MIN")
    assert torch.equal(m.psi_vjp(xb, wb1, wb2), tile)           # default switch: 50 001 walkers take the matrix-core path
    assert rel_l2(tile.cpu().numpy(), wave.cpu().numpy()) < 2e-4 and not torch.equal(tile, wave)
    # several chunks of a small workspace (the partial blocks of a two-row-block net are 8 512 floats each: 35 MB fixed for the four nets, + 256 B per walker)
    from waveflow_amd import _lib
    L = _lib.lib()
    ws = torch.empty(40 * 1024 * 1024, device="cuda", dtype=torch.uint8)    # room for ~ 19 000 walkers per chunk
    grad = torch.empty(m.n_params, device="cuda")
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
    rc = L.wf_psi_vjp(m._h, xb.data_ptr(), 50001, wb1.data_ptr(), wb2.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert not torch.equal(grad, wave) and rel_l2(grad.cpu().numpy(), tile.cpu().numpy()) < 1e-4      # (chunk sums in another order)
    grad2 = torch.empty_like(grad)
    assert L.wf_psi_vjp(m._h, xb.data_ptr(), 50001, wb1.data_ptr(), wb2.data_ptr(), grad2.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(grad, grad2)


def _leaves(tree):
    if isinstance(tree, (tuple, list)):
        return [a for t in tree for a in _leaves(t)]
    return [np.asarray(tree)]


@pytest.mark.gpu
def test_gradient_tile_path_other_models(monkeypatch):
    """The matrix-core gradient path on models the He checkpoint does not exercise (derivative boundary constraints, one and two layers, other
    boxes, degrees and knot counts), forced on 4 097 walkers against the reverse wave sweeps -- overall and LEAF BY LEAF (a wrong small leaf
    would hide behind the large ones), a non-zero boundary value on the prior included, and since round 4 the models with two row blocks per
    dimension (33 .. 64 bases: the 33-knot "32-bin" variant of C3); models outside its family (first-type box) keep the wave sweeps bit for bit."""
    import torch
    from waveflow_amd import checkpoint, flows, model_factory, wavefunctions
    mt = model_factory.get_masked_transform
    il, ir, pl, pr = {0: 0.0, 1: 0.0}, {0: 1.0, 1: 0.0}, {0: 0, 2: 0}, {0: 0, 1: 0}
    cases = [
        dict(L=3.0, n=2, k=6, kn=23, il=il, ir=ir, pl=pl, pr=pr, family=True),
        dict(L=2.0, n=1, k=5, kn=16, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}, family=True),
        dict(L=6.0, n=3, k=3, kn=10, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}, family=True),
        dict(L=10.0, n=3, k=6, kn=33, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}, family=True),      # 39 / 38 bases: two row blocks per dimension (round 4)
        dict(L=3.0, n=1, k=5, kn=16, il={0: 0.0}, ir={0: 1.0}, pl={0: 0.3}, pr={0: 0}, family=True),      # boundary value 0.3 on the prior (constant term)
        dict(L=3.0, n=2, k=5, kn=40, il=il, ir=ir, pl={0: 0.3, 2: 0}, pr=pr, family=True),                # two row blocks, derivative constraints, constant term
        dict(L=4.0, n=1, k=3, kn=58, il={0: 0.0}, ir={0: 1.0}, pl={0: 0}, pr={0: 0}, family=True),        # 61 / 60 bases: the second row block nearly full
    ]
    g = np.random.default_rng(17)
    for c in cases:
        init = wavefunctions.Waveflow(
            flows.Serial(flows.BoxTransformLayer(c["L"]), *(flows.IMADE(mt(), c["k"], c["kn"], 0.05, 1e-6, c["il"], c["ir"]), flows.Reverse()) * c["n"]),
            mt(allow_negative_params=True), c["k"], c["kn"], constraints_dict_left=c["pl"], constraints_dict_right=c["pr"],
            constrained_dimension_indices_left=[0], set_nn_output_grad_to_zero=False)
        params, psi, log_pdf, _ = init(4, 2)
        m = psi.model
        m.ensure_params(params)
        x = torch.as_tensor(sorted_walkers(4097, 2, 0.95 * c["L"], 13)).cuda()
        w1 = torch.as_tensor(g.normal(size=4097).astype(np.float32)).cuda()
        w2 = torch.as_tensor((0.05 * g.normal(size=4097)).astype(np.float32)).cuda()
        monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
        tile = m.psi_vjp(x, w1, w2)
        monkeypatch.setenv("WF_GRAD_TILE_MIN", "0")
        wave = m.psi_vjp(x, w1, w2)
        monkeypatch.delenv("WF_GRAD_TILE_MIN")
        if not c["family"]:
            assert torch.equal(tile, wave), c
            continue
        t, w = tile.cpu().numpy().astype(np.float64), wave.cpu().numpy().astype(np.float64)
        assert np.isfinite(t).all() and not np.array_equal(t, w), c
        assert rel_l2(t, w) < 2e-4, (c, rel_l2(t, w))
        scale = np.linalg.norm(w) / np.sqrt(w.size)
        for i, (lt, lw) in enumerate(zip(_leaves(checkpoint.unflatten_like(params, t)), _leaves(checkpoint.unflatten_like(params, w)))):
            if lw.size == 0:
                continue
            # a leaf's error against its own size, with a floor of 1e-3 of the gradient's rms entry (leaves that are zero by the masks stay zero)
            err = np.linalg.norm(lt - lw) / np.sqrt(lw.size)
            assert err <= 1e-3 * np.linalg.norm(lw) / np.sqrt(lw.size) + 1e-3 * scale * 1e-2, (c, i, lw.shape, err, np.linalg.norm(lw) / np.sqrt(lw.size))
            assert np.array_equal(lt == 0, lw == 0) or np.abs(lt[(lw == 0)]).max() <= 1e-5 * scale, (c, i)
    init = model_factory.get_waveflow_model(2, n_flow_layers=1, box_size=2, xu_coord_type="first")
    params, psi, log_pdf, _ = init(1, 2)
    m = psi.model
    m.ensure_params(params)
    x = torch.as_tensor(sorted_walkers(512, 2, 1.9, 3)).cuda()
    w = torch.ones(512, device="cuda")
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "1")
    tile = m.psi_vjp(x, w, w * 0.1)
    monkeypatch.setenv("WF_GRAD_TILE_MIN", "0")
    assert torch.equal(tile, m.psi_vjp(x, w, w * 0.1))


def test_psi_vjp_chunks_and_errors(he_flat):
    import torch
    from waveflow_amd import _lib
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    L = _lib.lib()
    x = torch.as_tensor(sorted_walkers(1000, 2, 8.0, 9)).cuda()
    w = torch.ones(1000, device="cuda")
    full = m.psi_vjp(x, w, w * 0.1)
    assert torch.equal(full, m.psi_vjp(x, w, w * 0.1))      # fixed-order reductions: bitwise reproducible
    # a workspace that only holds 128 samples forces 16 chunks: same gradient up to the grouping of the sums
    # (bytes per walker of the wave sweeps from a large batch: small ones are padded to the fixed part of the matrix-core path)
    per = L.wf_psi_vjp_workspace_bytes(m._h, 32768) // 32768 // 64
    ws = torch.empty(per * 128, device="cuda", dtype=torch.uint8)
    grad = torch.empty(m.n_params, device="cuda")
    w2 = (w * 0.1).contiguous()
    rc = L.wf_psi_vjp(m._h, x.data_ptr(), 1000, w.data_ptr(), w2.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_l2(grad.cpu().numpy(), full.cpu().numpy()) < 1e-4
    assert L.wf_psi_vjp(m._h, x.data_ptr(), 1000, w.data_ptr(), w2.data_ptr(), grad.data_ptr(), ws.data_ptr(), 16, None) == -1
    assert m.psi_vjp(np.zeros((0, 2), np.float32), np.zeros(0), np.zeros(0)).abs().sum().item() == 0.0


def test_training_reduces_the_energy_and_writes_the_reference_artefacts(tmp_path):
    import json
    from waveflow_amd import checkpoint, vqmc
    t = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=300, batch_size=512, log_every=150)
    t.save_dir = str(tmp_path / "He_1d_L10box")
    t.exact_sampler = True
    params, loss = t.start_training(verbose=False)
    l = np.asarray(loss[1:], dtype=np.float64)
    assert np.isfinite(l).all() and len(l) == 300
    assert np.median(l[-50:]) < 0.5 * np.median(l[:50])   # <E_L> falls (it starts around +10 .. +50 Ha; batch means are heavy-tailed)
    # the captured steps defer the evaluation tables of the large-batch kernel; they are current again when training returns
    xs = sorted_walkers(9000, 2, 9.0, 4)
    m = t.psi.model
    m.set_kernel("mfma")
    a = m.psi(xs)
    m.set_kernel("scalar")
    b = m.psi(xs)
    m.set_kernel("auto")
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-5 * np.abs(b).max())
    # artefacts of helpers.create_checkpoint_wavefunc / vqmc.py:79-88
    sd = t.save_dir
    assert json.load(open(f"{sd}/system_info.json"))["n_particle"] == 2
    p2, epoch = checkpoint.load_reference_checkpoint(f"{sd}/checkpoints")
    assert epoch == 300
    assert np.load(f"{sd}/outputs/wavefunctions_2d/values_epoch300.npy").shape == (10000,)
    assert np.load(f"{sd}/outputs/sample_points/values_epoch150.npy").shape == (250, 2)
    assert len(np.load(f"{sd}/loss.npy")) == 300
    # restart continues from the checkpoint (epoch counter, parameters and Adam's moments: optimizer_state.npz of the same epoch)
    opt = np.load(f"{sd}/optimizer_state.npz")
    assert int(opt["epoch"]) == 300 and opt["m"].shape == opt["v"].shape and np.abs(opt["m"]).max() > 0 and opt["v"].min() >= 0
    t2 = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=20, batch_size=512, log_every=10)
    t2.save_dir, t2.exact_sampler = sd, True
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # "moments restart from zero" would be raised here
        params2, loss2 = t2.start_training(restart=True, verbose=False)
    assert len(loss2) == len(np.load(f"{sd}/loss.npy")) + 1 or len(loss2) >= 320
    assert np.median(loss2[-20:]) < np.median(l[:50])
    assert checkpoint.load_reference_checkpoint(f"{sd}/checkpoints")[1] == 320


@pytest.mark.parametrize("knots", [23, 33])
def test_large_batch_training_steps_take_the_matrix_core_gradient(tmp_path, monkeypatch, knots):
    """From WF_GRAD_TILE_MIN walkers per step on, the captured training step refreshes every table and runs loss + gradient on the matrix cores
    (vqmc.py: _train_graphed); same seeds with the path switched off give the same loss curve up to the fp32 differences of the two kernels.
    33 knots: two row blocks per dimension (round 4)."""
    from waveflow_amd import vqmc
    curves = []
    for tm in ("16384", "0"):
        monkeypatch.setenv("WF_GRAD_TILE_MIN", tm)
        t = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=12, batch_size=16384, log_every=10 ** 9)
        t.num_knots = knots
        t.save_dir = str(tmp_path / f"He_tile_{tm}")
        t.exact_sampler = True
        params, loss = t.start_training(verbose=False)
        curves.append(np.asarray(loss[1:], dtype=np.float64))
    a, b = curves
    assert len(a) == 12 and np.isfinite(a).all() and np.isfinite(b).all()
    assert not np.array_equal(a, b)                       # two different kernels ...
    if knots == 23:
        # ... one training run (the walkers are the same: same sampler, same seeds): the two runs part at 2e-5 and drift (3.5e-3 at step 11 with round 4's draws)
        np.testing.assert_allclose(a[:2], b[:2], rtol=2e-4)
        np.testing.assert_allclose(a, b, rtol=2e-2)
    else:
        # the seeded 33-knot start is a rough one (batch means of E_L between 14 and 150 over these 12 steps): the two runs part at 1e-5 and
        # the differences grow with the steps (measured 1.4e-5, 2.3e-4, 1.7e-3, ... 3e-2 at step 11)
        np.testing.assert_allclose(a[:2], b[:2], rtol=2e-3)
        np.testing.assert_allclose(a, b, rtol=0.1)
        return
    # the staged sampler lifts the step's limit of 2^17 walkers (the wave sampler's): 2^18 walkers per step
    monkeypatch.delenv("WF_GRAD_TILE_MIN")
    t = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=3, batch_size=1 << 18, log_every=10 ** 9)
    t.save_dir = str(tmp_path / "He_2pow18")
    t.exact_sampler = True
    params, loss = t.start_training(verbose=False)
    assert len(loss[1:]) == 3 and np.isfinite(np.asarray(loss[1:], dtype=np.float64)).all()


def test_training_with_more_than_32_bases(tmp_path):
    """trainer.num_knots = 33 (examples/run_vqmc.py:10 pokes this attribute): 39 / 38 bases per dimension.  The sweeps run in the
    64-row layout, in the fused (captured) step like the 32-row ones."""
    from waveflow_amd import vqmc
    t = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=150, batch_size=256, log_every=10 ** 9)
    t.num_knots = 33
    t.save_dir = str(tmp_path / "He_33")
    t.exact_sampler = True
    params, loss = t.start_training(verbose=False)
    l = np.asarray(loss[1:], dtype=np.float64)
    assert (t.psi.model.i_nb, t.psi.model.p_nb) == (39, 38)
    assert np.isfinite(l).all() and len(l) == 150
    assert np.median(l[-30:]) < 0.5 * np.median(l[:30])


def test_train_step_and_train_step_uniform_vs_autograd_oracle(golden, he_flat):
    """vqmc.py:143-187: the two older training steps; the oracle differentiates the reference's loss expressions with autograd."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, vqmc
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    protons, _ = physics.system_catalogue[1]["He"]
    h_fn = physics.construct_hamiltonian_function(psi, protons=protons, n_space_dimensions=1, eps=0.0)
    mo = et.he_model(torch.float64)
    x = np.sort(golden["he_golden"]["sample_points"], -1)[:160].astype(np.float32)
    # E / psi and its derivative blow up where psi vanishes (loss_fn has no 1e-8 guard): keep walkers with a sizeable amplitude
    _, ps0, _ = et.hamiltonian(mo, he_flat, x.astype(np.float64), protons.reshape(-1))
    x = x[np.abs(ps0) > 0.05 * np.abs(ps0).max()]
    assert x.shape[0] > 64

    # --- train_step_uniform
    loss, grad = vqmc.loss_and_grad_uniform(params, psi, h_fn, x)
    lo, go = et.uniform_loss_grad(mo, he_flat, x.astype(np.float64), protons.reshape(-1))
    assert abs(loss - lo) < 1e-4 * max(1.0, abs(lo)), (loss, lo)
    assert abs(vqmc.loss_fn_uniform(params, psi, h_fn, x) - lo) < 1e-4 * max(1.0, abs(lo))
    rel_ok(grad.cpu().numpy().astype(np.float64), go, BOUND["L314"], "L314")

    # --- train_step
    g, loss = vqmc.train_step_gradients(params, psi, h_fn, log_pdf, x, running_average=-2.0)
    go, lo = et.train_step_gradients(mo, he_flat, x.astype(np.float64), protons.reshape(-1), -2.0)
    assert abs(loss - lo) < 2e-3 * max(1.0, abs(lo)), (loss, lo)
    g = g.cpu().numpy().astype(np.float64)
    assert np.abs(g).max() <= 10.0
    assert rel_l2(g, go) < 1e-2, rel_l2(g, go)
    gu, _ = vqmc.train_step_gradients(params, psi, h_fn, log_pdf, x, running_average=-2.0, clip=None)
    gou, _ = et.train_step_gradients(mo, he_flat, x.astype(np.float64), protons.reshape(-1), -2.0, clip=None)
    assert rel_l2(gu.cpu().numpy().astype(np.float64), gou) < 1e-2
    lv, (e, p) = vqmc.loss_fn(params, psi, h_fn, x)
    assert e.shape == p.shape == (x.shape[0], 1) and abs(lv - float((e / p).mean())) < 1e-6

    # --- the optimiser protocol: both steps move the parameters and return (state, loss)
    opt_init, opt_update, get_params = vqmc.adam(1e-3, model=psi.model)
    st = opt_init(params)
    before = flatten_params(params).copy()
    st, l1 = vqmc.train_step(0, psi, h_fn, log_pdf, opt_update, st, get_params, x, -2.0)
    st, l2 = vqmc.train_step_uniform(1, psi, h_fn, opt_update, st, get_params, x)
    after = st.x.cpu().numpy()
    assert np.isfinite(l1) and np.isfinite(l2) and np.isfinite(after).all()
    moved = np.abs(after - before)
    assert 0 < moved.max() < 5e-3     # two Adam steps of size 1e-3


def test_loss_fn_efficient_value_matches_sums(he_flat):
    from waveflow_amd import vqmc
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    protons, _ = physics.system_catalogue[1]["He"]
    h_fn = physics.construct_hamiltonian_function(psi, protons=protons, n_space_dimensions=1, eps=0.0)
    x = sorted_walkers(300, 2, 6.0, 11)
    lv = vqmc.loss_fn_efficient(params, psi, h_fn, x, 0.0)
    loss, grad, (mean, var, se) = vqmc.loss_and_grad_efficient(params, psi, h_fn, x, 0.0)
    # psi comes from the MFMA kernel in loss_fn_efficient and from the energy kernel in the fused path: near the node the
    # ratio H psi / psi amplifies their fp32 differences
    assert abs(lv - loss) < 1e-2 * max(1.0, abs(lv)), (lv, loss)
    assert abs(mean - loss) < 1e-12
    grad = grad.cpu().numpy()
    assert grad.shape == (he_flat.size,) and np.isfinite(grad).all()


def test_logpdf_vjp_waveflow_vs_autograd_oracle(golden, he_flat):
    import torch
    from oracle import energy_torch as et
    params, psi, log_pdf, sample = he(he_flat)
    m = log_pdf.model
    m.ensure_params(params)
    x = np.concatenate([np.sort(golden["he_golden"]["sample_points"], -1)[:96], sorted_walkers(32, 2, 8.0, 6)]).astype(np.float32)
    w = np.random.default_rng(4).normal(size=128).astype(np.float32)
    got = m.logpdf_vjp(x, w).cpu().numpy().astype(np.float64)
    want = et.logpdf_vjp(et.he_model(torch.float64), he_flat, x.astype(np.float64), w)
    assert ((want == 0) <= (np.abs(got) <= 1e-12)).all()
    assert rel_l2(got, want) < 1e-3, rel_l2(got, want)


def _directional_check(log_pdf, params, om, X, seed, n_dirs=6, tol=2e-3):
    """sum_b w_b log_pdf_b along random parameter directions: HIP gradient . d vs central differences of the fp64 C oracle."""
    from waveflow_amd import flatten_params
    m = log_pdf.model
    m.ensure_params(params)
    flat = flatten_params(params)
    g = np.random.default_rng(seed)
    w = g.normal(size=X.shape[0]).astype(np.float32)
    grad = m.logpdf_vjp(X, w).cpu().numpy().astype(np.float64)
    assert grad.shape == flat.shape and np.isfinite(grad).all() and np.abs(grad).max() > 0
    F = lambda f: float((w.astype(np.float64) * om.log_pdf(f.astype(np.float32), X, f64=True)).sum())
    worst = 0.0
    for i in range(n_dirs):
        # random directions, and directions confined to one leaf block so that every weight group is probed
        d = g.normal(size=flat.size)
        if i >= 2:
            lo = g.integers(0, flat.size - 64)
            mask = np.zeros(flat.size)
            mask[lo:lo + g.integers(16, 4096)] = 1
            d = d * mask
        d = (d / np.linalg.norm(d)).astype(np.float32)
        eps = 2e-3
        fd = (F(flat + eps * d) - F(flat - eps * d)) / (2 * eps)
        an = float(grad @ d.astype(np.float64))
        scale = max(abs(fd), 1e-2 * np.linalg.norm(grad) / np.sqrt(flat.size) * 10, 1e-6)
        worst = max(worst, abs(fd - an) / scale)
        assert abs(fd - an) <= tol * scale + 1e-4 * np.linalg.norm(grad), (i, fd, an)
    return worst


def test_logpdf_vjp_benchmark_models_vs_fp64_finite_differences():
    """benchmark_tests.get_model's three families (benchmark_tests.py:50-79): MFlow, IFlow, Flow."""
    import os
    import oracle
    from conftest import GOLDEN
    from waveflow_amd import flows, model_factory
    X = np.load(os.path.join(GOLDEN, "circles_x256.npy")).astype(np.float32)[:128]
    mt = model_factory.get_masked_transform
    init = flows.MFlow(flows.Serial(*(flows.IMADE(mt(), spline_degree=5, n_internal_knots=15, spline_regularization=0.01,
                                                  reverse_fun_tol=1e-6), flows.Reverse()) * 3), mt(), spline_degree=3, n_internal_knots=15)
    params, log_pdf, _ = init(0, 2)
    om = oracle.Model(D=2, n_layers=3, i_k=5, i_knots=15, i_reg=0.01, prior="mflow", p_k=3, p_knots=15)
    _directional_check(log_pdf, params, om, X, 1)
    init = flows.Flow(flows.Serial(*(flows.IMADE(mt(), spline_degree=5, n_internal_knots=15, spline_regularization=0.1,
                                                 reverse_fun_tol=1e-6), flows.Reverse()) * 2), flows.Uniform(), prior_support=(0.0, 1.0))
    params, log_pdf, _ = init(2, 2)
    om = oracle.Model(D=2, n_layers=2, i_k=5, i_knots=15, i_reg=0.1, prior="uniform")
    _directional_check(log_pdf, params, om, X, 2)
    init = flows.Flow(flows.Serial(*(flows.MADE(mt(return_simple_masked_transform=True)), flows.Reverse()) * 3), flows.Normal(-0.5))
    params, log_pdf, _ = init(3, 2)
    om = oracle.Model(D=2, n_layers=3, layer_kind="made", prior="normal", normal_offset=-0.5)
    _directional_check(log_pdf, params, om, X, 3)


@pytest.mark.parametrize("model_type", ["Flow", "IFlow", "MFlow"])
def test_benchmark_training_lowers_the_negative_log_likelihood(tmp_path, model_type):
    """benchmark_tests.train_model on the committed 256 circles points: loss falls, artefacts follow helpers.py:176-214."""
    import os
    from conftest import GOLDEN
    from waveflow_amd import benchmark_tests
    X = np.load(os.path.join(GOLDEN, "circles_x256.npy")).astype(np.float32)
    params, losses = benchmark_tests.train_model(X, 120, 500, model_type=model_type, dataset_name="circles", check_step=60,
                                                 spline_reg=0.01, save_dir=str(tmp_path), ngrid=40, num_flow_layer=2, spline_degree=5,
                                                 num_knots=15, step_size=2e-3, verbose=False)
    assert len(losses) == 121 and np.isfinite(losses).all()
    assert losses[-1] < losses[0] - 0.05, (losses[0], losses[-1])
    run = [p for p in (tmp_path / "circles").iterdir()][0]
    assert np.load(run / "outputs" / "pdf_grid_epoch60.npy").shape == (40, 40)
    assert np.load(run / "outputs" / "samples_epoch120.npy").shape == (500, 2)
    assert len(np.loadtxt(run / "kl_divergences.txt")) == 3 and len(np.loadtxt(run / "losses.txt")) >= 120
    assert (run / "system_info.json").exists()


@pytest.mark.parametrize("D,box,layers,k,kn,B", [(3, "first", 2, 4, 13, 37), (4, "mean", 1, 5, 16, 65), (2, "first", 1, 3, 10, 1), (8, "mean", 2, 4, 13, 9),
                                              (5, "first", 1, 3, 10, 6), (2, "mean", 2, 6, 33, 33), (3, "first", 1, 5, 30, 17), (4, "mean", 1, 4, 41, 5)])
def test_gradients_other_shapes_vs_autograd_oracle(D, box, layers, k, kn, B):
    """D = 3 .. 8 (C4: the 8-electron chain), both box transforms, ragged batches, 33..64 bases per dimension (the 33-knot
    "32-bin" variant of C3: one dimension x 64 rows per output pass): psi / Laplacian / log_pdf gradients and the local energy
    vs torch (fp64)."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, model_factory
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=k, i_spline_degree=k, n_prior_internal_knots=kn, n_i_internal_knots=kn,
                                                i_spline_reg=0.05, n_flow_layers=layers, box_size=5.0, xu_coord_type=box)
    params, psi, log_pdf, sample = init_fun(7, D)
    m = psi.model
    m.ensure_params(params)
    flat = flatten_params(params)
    constr = tuple(range(0, D - 1)) if box == "mean" else tuple(range(1, D))
    mo = et.TorchWaveflow(D, layers, box, 5.0, k, kn, 0.05, constr, dtype=torch.float64)
    x = sorted_walkers(B, D, 4.5, 21)
    g = np.random.default_rng(8)
    wp, wl, w = (g.normal(size=B).astype(np.float32) for _ in range(3))
    got = m.psi_vjp(x, wp, wl).cpu().numpy().astype(np.float64)
    want = et.psi_vjp(mo, flat, x.astype(np.float64), wp, wl)
    assert rel_l2(got, want) < 3e-3, rel_l2(got, want)
    got = m.logpdf_vjp(x, w).cpu().numpy().astype(np.float64)
    want = et.logpdf_vjp(mo, flat, x.astype(np.float64), w)
    assert rel_l2(got, want) < 3e-3, rel_l2(got, want)
    protons = np.linspace(-3.0, 3.0, 4)
    hp, ps, lap = m.hamiltonian(x, protons, return_psi=True, return_laplacian=True)
    ho, po, lo = et.hamiltonian(mo, flat, x.astype(np.float64), protons)
    np.testing.assert_allclose(ps, po, rtol=2e-4, atol=1e-6 * np.abs(po).max())
    np.testing.assert_allclose(lap, lo, rtol=0, atol=2e-3 * np.abs(lo).max())
    np.testing.assert_allclose(hp, ho, rtol=0, atol=2e-3 * np.abs(ho).max())


def test_device_parameter_path_matches_host_path(he_flat):
    """wf_model_set_params_device fills the same images as wf_model_set_params; wf_adam_step follows the host Adam."""
    import torch
    from waveflow_amd import DeviceParams, vqmc
    params, psi, log_pdf, sample = he(he_flat)
    x = sorted_walkers(4096, 2, 9.0, 3)
    want_lp, want_psi = log_pdf(params, x), psi(params, x)
    other = he_flat + np.float32(1e-3)
    log_pdf.model.set_params(other)                       # overwrite, then come back through the device path
    dp = DeviceParams(params, torch.as_tensor(he_flat).cuda())
    assert np.array_equal(log_pdf(dp, x), want_lp) and np.array_equal(psi(dp, x), want_psi)
    # both kernels read images written by k_pack: the scalar kernel too
    log_pdf.model.set_kernel("scalar")
    got = log_pdf(dp, x)
    log_pdf.model.set_params(he_flat)
    assert np.array_equal(got, log_pdf.model.log_pdf(x))
    log_pdf.model.set_kernel("auto")
    tree = dp.tree()
    assert np.array_equal(np.concatenate([np.ravel(a) for a in vqmc.flatten_params(tree)[None]]), he_flat)
    # Adam: device state vs host state over a few steps with the same gradients
    g = np.random.default_rng(0)
    oi_h, ou_h, gp_h = vqmc.adam(1e-2)
    oi_d, ou_d, gp_d = vqmc.adam(1e-2, model=log_pdf.model)
    sh, sd = oi_h(params), oi_d(params)
    for i in range(12):
        gr = g.normal(size=he_flat.size).astype(np.float32)
        sh = ou_h(i, gr, sh)
        sd = ou_d(i, torch.as_tensor(gr).cuda(), sd)
    np.testing.assert_allclose(gp_d(sd).flat.cpu().numpy(), sh.x, rtol=0, atol=2e-6)
    assert isinstance(gp_d(sd), DeviceParams) and gp_d(sd).version == 12


def test_training_step_is_graph_capturable(he_flat):
    """loss + gradient, Adam and the image refill run without host work: capture them once, replay, compare with eager."""
    import torch
    from waveflow_amd import DeviceParams
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    protons = physics.system_catalogue[1]["He"][0].reshape(-1)
    x = torch.as_tensor(sorted_walkers(512, 2, 6.0, 13)).cuda()

    def fresh():
        xs = torch.as_tensor(he_flat).cuda()
        return xs, torch.zeros_like(xs), torch.zeros_like(xs)

    def step(xs, mm, vv, i):
        m.set_params_device(xs)
        sums, grad = m.vqmc_loss_grad(x, protons, -2.0)
        m.adam_step(xs, grad, mm, vv, i, 1e-3)
        return sums

    # eager reference: three steps
    xe, me, ve = fresh()
    se = [step(xe, me, ve, i).clone() for i in range(3)]
    torch.cuda.synchronize()
    # captured: the step index only enters through the bias corrections, so capture step 0..2 as three graphs' worth of
    # launches in one graph
    xg, mg, vg = fresh()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(xg, mg, vg, 0)          # warm-up on the capture stream (workspace allocation happens here)
        xg.copy_(torch.as_tensor(he_flat)); mg.zero_(); vg.zero_()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            sg = [step(xg, mg, vg, i).clone() for i in range(3)]
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    # every reduction runs in a fixed order (no atomics): the replay reproduces the eager run bit for bit
    assert np.array_equal(xg.cpu().numpy(), xe.cpu().numpy())
    for a_, b_ in zip(sg, se):
        assert np.array_equal(a_.cpu().numpy(), b_.cpu().numpy())
    assert np.abs(xg.cpu().numpy() - he_flat).max() > 1e-3


def test_run_vqmc_example_under_torchrun_single_rank(tmp_path):
    """examples/run_vqmc.py through torch.distributed.run with one rank: RCCL initialises, the packed all-reduce path runs."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, WF_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "examples", "run_vqmc.py"), "--epochs", "40", "--batch", "256", "--lr", "1e-3",
           "--log-every", "20", "--exact-sampler", "--save-dir", str(tmp_path / "run")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.load(open(tmp_path / "run" / "system_info.json"))["system_name"] == "He"
    loss = np.load(tmp_path / "run" / "loss.npy")
    assert len(loss) >= 39 and np.isfinite(loss).all()


def test_two_rank_training_keeps_replicas_identical(tmp_path):
    """Two ranks (both on cuda:0, gloo) split every step's walkers: after the packed all-reduce both replicas hold the same
    parameters bit for bit, see the same global loss, and rank 0 alone wrote the artefacts."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_two_rank_train.py"), str(tmp_path), "25", "300"]
    r = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    p0, p1 = np.load(tmp_path / "params_rank0.npy"), np.load(tmp_path / "params_rank1.npy")
    l0, l1 = np.load(tmp_path / "loss_rank0.npy"), np.load(tmp_path / "loss_rank1.npy")
    assert np.array_equal(p0, p1) and np.array_equal(l0, l1)
    assert np.isfinite(p0).all() and np.isfinite(l0).all() and len(l0) == 25
    assert (tmp_path / "run" / "checkpoints").exists()
    # the walkers really are sharded: a single process with the same settings draws all 300 walkers of a step from ONE sampler
    # stream, the two ranks draw 150 each from their own streams, so the loss traces differ (round 1 passed the default process
    # group on as "no group" and every rank silently trained the whole batch: the traces were then identical)
    single = tmp_path / "single"
    single.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_two_rank_train.py"), str(single), "25", "300"],
                       env=dict(os.environ, WF_TEST_BACKEND="none"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ls = np.load(single / "loss_rank0.npy")
    assert len(ls) == 25 and not np.array_equal(ls, l0)
    assert abs(np.mean(ls) - np.mean(l0)) < 5 * (np.std(ls) + np.std(l0)) / np.sqrt(25) + 1.0


def test_rccl_single_rank_training_and_bench(tmp_path):
    """RCCL itself (backend "nccl"), with the one rank this box has: the sharded training step wf_vqmc_train_step_local -> all-reduce of
    the packed [gradient, sum E, sum E^2, n] buffer -> wf_vqmc_train_step_apply, issued call by call and captured in a hipGraph with the
    collective inside (WF_GRAPH_COLLECTIVE=1), and bench.py's distributed path (WF_FORCE_DIST=1)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT

    def port():
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        p = sock.getsockname()[1]
        sock.close()
        return p

    out = {}
    for gc in ("0", "1"):
        d = tmp_path / f"gc{gc}"
        d.mkdir()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port()), os.path.join(ROOT, "tests", "_two_rank_train.py"), str(d), "25", "256"]
        env = dict(os.environ, WF_TEST_BACKEND="nccl", WF_GRAPH_COLLECTIVE=gc, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0 and gc == "1" and "SIGABRT" in r.stderr:
            # The opt-in path (an RCCL collective inside a captured graph) aborted once in ~5 suite runs on this pool, inside the child at an
            # unknown point (torchrun reports "Signal 6 (SIGABRT)"), and passes when run again; the default path (gc == "0") never did.  One retry,
            # with the first failure printed, so that a reproducible abort still fails the test.
            print("WF_GRAPH_COLLECTIVE=1 run aborted, retrying once:\n" + r.stderr[-1500:])
            r = subprocess.run(cmd[:9] + [str(port())] + cmd[10:], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, f"WF_GRAPH_COLLECTIVE={gc}: " + r.stdout[-2000:] + r.stderr[-3000:]
        out[gc] = (np.load(d / "params_rank0.npy"), np.load(d / "loss_rank0.npy"))
        assert np.isfinite(out[gc][0]).all() and np.isfinite(out[gc][1]).all() and len(out[gc][1]) == 25
    # the captured sequence (collective included) replays the same arithmetic as the three calls
    assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--batch", "65536",
                        "--no-cpu-baseline", "--no-extras"],
                       env=dict(os.environ, WF_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port()), HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["rccl_ranks"] == 1 and d["dist_backend"] == "nccl" and d["n_gpus"] == 1


def test_abi_error_paths_of_the_gradient_entry_points(he_flat):
    import torch
    from waveflow_amd import _lib, flows, model_factory
    L = _lib.lib()
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    m.ensure_params(params)
    x = torch.as_tensor(sorted_walkers(8, 2, 5.0, 1)).cuda()
    w = torch.ones(8, device="cuda")
    g = torch.empty(m.n_params, device="cuda")
    ws = torch.empty(int(L.wf_psi_vjp_workspace_bytes(m._h, 8)), device="cuda", dtype=torch.uint8)
    # null pointers / wrong sizes
    assert L.wf_psi_vjp(m._h, x.data_ptr(), 8, None, w.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), None) == -1
    assert L.wf_logpdf_vjp(m._h, x.data_ptr(), 8, w.data_ptr(), None, ws.data_ptr(), ws.numel(), None) == -1
    assert L.wf_vqmc_loss_grad(m._h, x.data_ptr(), 8, None, 9, 0.0, 1.0, w.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), None) == -1
    assert L.wf_model_set_params_device(m._h, g.data_ptr(), 5, None) == -1
    assert L.wf_adam_step(None, g.data_ptr(), g.data_ptr(), g.data_ptr(), 4, 0, 1e-3, 0.9, 0.999, 1e-8, None) == -1
    assert L.wf_psi_vjp_workspace_bytes(None, 8) == -1
    # empty batches are fine and give a zero gradient
    assert L.wf_vqmc_loss_grad(m._h, None, 0, None, 0, 0.0, 1.0, None, g.data_ptr(), None, 0, None) == 0
    torch.cuda.synchronize()
    assert g.abs().sum().item() == 0.0
    # psi / Laplacian gradients need the Waveflow prior; log_pdf gradients work for the density models
    p2, lp2, _ = model_factory.get_model(n_flow_layers=1, prior_constraint_dict_left={0: 0}, prior_constraint_dict_right={0: 0},
                                         i_constraint_dict_left={0: 0.0}, i_constraint_dict_right={0: 1.0})(0, 2)
    lp2.model.ensure_params(p2)
    assert L.wf_psi_vjp_workspace_bytes(lp2.model._h, 8) == -2
    xs = torch.rand(8, 2, device="cuda") * 0.8 + 0.1
    assert torch.isfinite(lp2.model.logpdf_vjp(xs, w)).all()
    # general homogeneous constraint dictionaries (derivative orders > 0, every value 0 -- tests/test_boundary_constraints.py:30-31 style):
    # the table-driven kernels carry the boundary map in their tables, gradients included (fp64 finite differences of the C oracle)
    import oracle
    p3, lp3, _ = model_factory.get_model(n_flow_layers=1, i_constraint_dict_left={0: 0.0, 2: 0.0, 3: 0.0}, i_constraint_dict_right={0: 1.0},
                                         prior_constraint_dict_left={0: 0, 2: 0})(0, 2)
    om3 = oracle.Model(D=2, n_layers=1, i_k=5, i_knots=15, i_reg=0.0, i_left={0: 0.0, 2: 0.0, 3: 0.0}, i_right={0: 1.0}, prior="mflow", p_k=5,
                       p_knots=15, p_left={0: 0.0, 2: 0.0}, p_right={})
    X3 = (np.random.default_rng(8).random((96, 2)) * 0.9 + 0.05).astype(np.float32)
    _directional_check(lp3, p3, om3, X3, seed=12)
    # the same density model with gated heads (set_nn_output_grad_to_zero=True: the exact configuration of tests/test_boundary_constraints.py:30-31):
    # sigmoid heads, |zero_params| -- the gradient of a leaf entry carries the sign of the raw value
    p5, lp5, _ = model_factory.get_model(n_flow_layers=1, i_constraint_dict_left={0: 0.0, 2: 0.0, 3: 0.0}, i_constraint_dict_right={0: 1.0},
                                         prior_constraint_dict_left={0: 0, 2: 0}, set_nn_output_grad_to_zero=True)(0, 2)
    om5 = oracle.Model(D=2, n_layers=1, i_k=5, i_knots=15, i_reg=0.0, i_left={0: 0.0, 2: 0.0, 3: 0.0}, i_right={0: 1.0}, prior="mflow", p_k=5,
                       p_knots=15, p_left={0: 0.0, 2: 0.0}, p_right={}, i_gate=True, p_gate=True)
    _directional_check(lp5, p5, om5, X3, seed=13)
    # a constraint with a non-zero value adds a constant term b to the map.  I- and M-spline coefficients reach the constraints normalised
    # (sum 1), so the table-driven kernels fold b into the linear part (A + b 1^T: wf_model.cpp bc_map), gradients included
    p4, lp4, _ = model_factory.get_model(n_flow_layers=1, i_constraint_dict_left={0: 0.0, 1: 0.5}, i_constraint_dict_right={0: 1.0, 1: 0.25},
                                         prior_constraint_dict_left={0: 0.3})(0, 2)
    om4 = oracle.Model(D=2, n_layers=1, i_k=5, i_knots=15, i_reg=0.0, i_left={0: 0.0, 1: 0.5}, i_right={0: 1.0, 1: 0.25}, prior="mflow", p_k=5,
                       p_knots=15, p_left={0: 0.3}, p_right={})
    _directional_check(lp4, p4, om4, X3, seed=14)
    # the B-spline prior's coefficients reach the constraints divided by their signed sum and are normalised afterwards: its constant term
    # travels as a separate vector (round 3: c = (A o) @ ob_to_b + (sum o) (b @ ob_to_b)) -- table-driven kernels and gradients cover it
    # (test_wavefunction_with_a_nonzero_boundary_value_on_the_prior); here: the flow-free model of the reference's boundary test
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, flows, wavefunctions
    mt = model_factory.get_masked_transform
    init6 = wavefunctions.Waveflow(flows.Serial(flows.BoxTransformLayer(1.0)), mt(allow_negative_params=True), 5, 16,
                                   constraints_dict_left={0: 0.1}, constraints_dict_right={0: 0}, constrained_dimension_indices_left=[0],
                                   set_nn_output_grad_to_zero=False)
    p6, psi6, lp6, _ = init6(0, 2)
    lp6.model.ensure_params(p6)
    x6 = sorted_walkers(8, 2, 1.0, 3)
    mo6 = et.TorchWaveflow(2, 0, "mean", 1.0, 5, 16, 0.0, (0,), dtype=torch.float64, p_left={0: 0.1}, p_right={0: 0.0})
    f6 = flatten_params(p6)
    for kernel in ("scalar", "mfma", "wave"):
        lp6.model.set_kernel(kernel)
        np.testing.assert_allclose(lp6(p6, x6), mo6.log_pdf(f6, torch.as_tensor(x6, dtype=torch.float64)).numpy(), rtol=0, atol=2e-3, err_msg=kernel)
    lp6.model.set_kernel("auto")
    got = lp6.model.logpdf_vjp(torch.as_tensor(x6).cuda(), w).cpu().numpy().astype(np.float64)      # (w: eight ones)
    want = et.logpdf_vjp(mo6, f6, x6.astype(np.float64), np.ones(8, np.float32))
    assert rel_l2(got, want) < 2e-3, rel_l2(got, want)


def test_model_without_flow_layers(he_flat):
    """get_waveflow_model(input_dim, n_flow_layers=0) -- the configuration of the reference's tests/test_boundary_constraints.py:
    box transform + prior only.  psi, log_pdf, H psi and the gradients vs the torch oracle; sampler round trip."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, model_factory
    init_fun = model_factory.get_waveflow_model(2, n_flow_layers=0)          # reference defaults: k = 5, 16 knots, box_size 1
    params, psi, log_pdf, sample = init_fun(3, 2)
    flat = flatten_params(params)
    mo = et.TorchWaveflow(2, 0, "mean", 1.0, 5, 16, 0.0, (0,), dtype=torch.float64)
    g = np.linspace(0.01, 0.99, 20)
    x = np.sort(np.stack([g * 1.6 - 0.8, g[::-1] * 0.9 - 0.1], -1), -1).astype(np.float32)
    for kernel in ("scalar", "mfma", "wave"):
        try:
            psi.model.set_kernel(kernel)
        except Exception:
            continue
        np.testing.assert_allclose(psi(params, x), mo.psi(flat, torch.as_tensor(x, dtype=torch.float64)).numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(log_pdf(params, x), mo.log_pdf(flat, torch.as_tensor(x, dtype=torch.float64)).numpy(), rtol=0, atol=2e-3)
    psi.model.set_kernel("auto")
    hp, ps, lap = psi.model.hamiltonian(x, [0.0], return_psi=True, return_laplacian=True)
    ho, po, lo = et.hamiltonian(mo, flat, x.astype(np.float64), [0.0])
    np.testing.assert_allclose(lap, lo, rtol=0, atol=2e-3 * np.abs(lo).max())
    w = np.ones(20, np.float32)
    got = psi.model.psi_vjp(x, w, 0.1 * w).cpu().numpy().astype(np.float64)
    want = et.psi_vjp(mo, flat, x.astype(np.float64), w, 0.1 * w)
    assert rel_l2(got, want) < 3e-3
    s, lat = sample(5, params, 200, return_original_samples=True, exact_inverse=True)
    lp, u = log_pdf(params, s, return_sample=True)
    assert (u - lat).abs().max().item() < 1e-4 and bool(torch.isfinite(lp).all())


def test_wavefunction_with_derivative_constraints_energy_and_gradients():
    """General homogeneous boundary dictionaries through the wave sweeps: psi, H psi, the Laplacian and both vector-Jacobian products of a
    Waveflow whose layers and prior carry derivative constraints ({0: 0, 1: 0} / {0: 1, 1: 0}; {0: 0, 2: 0} / {0: 0, 1: 0}), vs the torch
    oracle that enforces the dictionaries literally (sequential overwrite, as the reference does) and differentiates with autograd."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, flows, model_factory, wavefunctions
    mt = model_factory.get_masked_transform
    il, ir, pl, pr = {0: 0.0, 1: 0.0}, {0: 1.0, 1: 0.0}, {0: 0, 2: 0}, {0: 0, 1: 0}
    init = wavefunctions.Waveflow(
        flows.Serial(flows.BoxTransformLayer(3.0), *(flows.IMADE(mt(), 6, 23, 0.05, 1e-6, il, ir), flows.Reverse()) * 2),
        mt(allow_negative_params=True), 6, 23, constraints_dict_left=pl, constraints_dict_right=pr, constrained_dimension_indices_left=[0],
        set_nn_output_grad_to_zero=False)
    params, psi, log_pdf, _ = init(4, 2)
    flat = flatten_params(params)
    mo = et.TorchWaveflow(2, 2, "mean", 3.0, 6, 23, 0.05, (0,), dtype=torch.float64, i_left=il, i_right=ir, p_left=pl, p_right=pr)
    x = sorted_walkers(160, 2, 2.7, 13)
    m = psi.model
    m.ensure_params(params)
    assert m.desc.i_left.n == 2 and m.desc.p_left.n == 2
    xt = torch.as_tensor(x, dtype=torch.float64)
    for kernel in ("scalar", "mfma", "wave"):
        m.set_kernel(kernel)
        np.testing.assert_allclose(psi(params, x), mo.psi(flat, xt).numpy(), rtol=0, atol=3e-5 * float(mo.psi(flat, xt).abs().max()))
        np.testing.assert_allclose(log_pdf(params, x), mo.log_pdf(flat, xt).numpy(), rtol=0, atol=3e-3)
    m.set_kernel("auto")
    hp, ps, lap = m.hamiltonian(x, [0.0, 0.0], return_psi=True, return_laplacian=True)
    ho, po, lo = et.hamiltonian(mo, flat, x.astype(np.float64), [0.0, 0.0])
    np.testing.assert_allclose(lap, lo, rtol=0, atol=3e-3 * np.abs(lo).max())
    np.testing.assert_allclose(hp, ho, rtol=0, atol=3e-3 * np.abs(ho).max())
    g = np.random.default_rng(2)
    w1, w2 = g.normal(size=len(x)).astype(np.float32), g.normal(size=len(x)).astype(np.float32)
    got = m.psi_vjp(x, w1, 0.1 * w2).cpu().numpy().astype(np.float64)
    want = et.psi_vjp(mo, flat, x.astype(np.float64), w1, 0.1 * w2)
    rel_ok(got, want, BOUND["L803"], "L803")
    got = m.logpdf_vjp(x, w1).cpu().numpy().astype(np.float64)
    want = et.logpdf_vjp(mo, flat, x.astype(np.float64), w1)
    assert rel_l2(got, want) < 2e-3, rel_l2(got, want)


@pytest.mark.parametrize("D", [2, 3])
def test_wavefunction_with_a_nonzero_boundary_value_on_the_prior(D):
    """A boundary dictionary with a NON-ZERO value on the B-spline prior (bsplines_jax.py:173-199: value 0.2 and first derivative 8 at the left end, -0.1 / -5 at the right one):
    the net's outputs reach the constraints divided by their signed sum S (model_factory.py:69), w' = (A o + S b) / S, so the table-driven
    kernels carry the constant term as c = (A o) @ ob_to_b + S (b @ ob_to_b) -- k_mfma (accumulator update), the wave sweeps (psi, H psi,
    Laplacian) and the reverse sweeps (both vector-Jacobian products: the term reaches every raw output through S and the weights through
    the norm).  Against the torch oracle, which enforces the dictionaries literally, and the per-walker kernel's literal sequence."""
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, flows, model_factory, wavefunctions
    mt = model_factory.get_masked_transform
    il, ir, pl, pr = {0: 0.0}, {0: 1.0}, {0: 0.2, 1: 8.0}, {0: -0.1, 1: -5.0}
    init = wavefunctions.Waveflow(
        flows.Serial(flows.BoxTransformLayer(3.0), *(flows.IMADE(mt(), 6, 23, 0.05, 1e-6, il, ir), flows.Reverse()) * 2),
        mt(allow_negative_params=True), 6, 23, constraints_dict_left=pl, constraints_dict_right=pr, constrained_dimension_indices_left=list(range(D - 1)),
        set_nn_output_grad_to_zero=False)
    params, psi, log_pdf, _ = init(5, D)
    flat = flatten_params(params)
    mo = et.TorchWaveflow(D, 2, "mean", 3.0, 6, 23, 0.05, tuple(range(D - 1)), dtype=torch.float64, i_left=il, i_right=ir, p_left=pl, p_right=pr)
    mo0 = et.TorchWaveflow(D, 2, "mean", 3.0, 6, 23, 0.05, tuple(range(D - 1)), dtype=torch.float64, i_left=il, i_right=ir, p_left={0: 0.0, 1: 0.0},
                           p_right={0: 0.0, 1: 0.0})
    x = sorted_walkers(160, D, 2.7, 13)
    xt = torch.as_tensor(x, dtype=torch.float64)
    m = psi.model
    m.ensure_params(params)
    po = mo.psi(flat, xt).numpy()
    assert np.abs(po - mo0.psi(flat, xt).numpy()).max() > 2e-2 * np.abs(po).max()      # the constant term matters on this model
    for kernel in ("scalar", "mfma", "wave"):
        m.set_kernel(kernel)     # (no skip: every kernel covers the dictionary)
        np.testing.assert_allclose(psi(params, x), po, rtol=0, atol=3e-5 * float(np.abs(po).max()), err_msg=kernel)
        # (log of psi^2 + 1e-7 near the nodes: fp32 against the fp64 oracle; the literal per-walker kernel misses 3e-3 on one D = 3 walker at -18)
        np.testing.assert_allclose(log_pdf(params, x), mo.log_pdf(flat, xt).numpy(), rtol=3e-4, atol=3e-3, err_msg=kernel)
    m.set_kernel("auto")
    # a large batch goes through k_mfma (the composite dimension-0 tables carry the term too): against the wave kernel
    xb = sorted_walkers(20000, D, 2.7, 14)
    m.set_kernel("mfma"); p_m = psi(params, xb)
    m.set_kernel("wave"); p_w = psi(params, xb)
    m.set_kernel("auto")
    assert np.abs(p_m - p_w).max() <= 5e-5 * np.abs(p_w).max()
    hp, ps, lap = m.hamiltonian(x, [0.0] * D, return_psi=True, return_laplacian=True)
    ho, _, lo = et.hamiltonian(mo, flat, x.astype(np.float64), [0.0] * D)
    np.testing.assert_allclose(lap, lo, rtol=0, atol=3e-3 * np.abs(lo).max())
    np.testing.assert_allclose(hp, ho, rtol=0, atol=3e-3 * np.abs(ho).max())
    # 20 000 walkers: with two particles H psi goes through the one-kernel matrix-core form, which carries the term channel by channel
    # (c += (sum o) (b @ ob_to_b) for the value and both derivative channels); for D = 3 it stays on the wave sweeps.  Against the wave sweeps:
    hb, pb, lb = (np.asarray(t) for t in m.hamiltonian(xb, [0.0] * D, return_psi=True, return_laplacian=True))
    assert np.isfinite(hb).all() and np.abs(pb - p_w).max() <= 5e-5 * np.abs(p_w).max()
    import os
    os.environ["WF_ENERGY_TILE_MIN"] = "0"
    try:
        hw, pw2, lw = (np.asarray(t) for t in m.hamiltonian(xb, [0.0] * D, return_psi=True, return_laplacian=True))
    finally:
        del os.environ["WF_ENERGY_TILE_MIN"]
    if D == 2:
        assert not np.array_equal(hb, hw)                       # two different kernels
    for a_, b_ in ((pb, pw2), (lb, lw), (hb, hw)):
        d_ = np.abs(a_ - b_)
        assert d_.max() <= 5e-4 * np.abs(b_).max() and np.median(d_) <= 1e-6 * np.abs(b_).max(), (d_.max() / np.abs(b_).max(), np.median(d_) / np.abs(b_).max())
    g = np.random.default_rng(2)
    w1, w2 = g.normal(size=len(x)).astype(np.float32), g.normal(size=len(x)).astype(np.float32)
    got = m.psi_vjp(x, w1, 0.1 * w2).cpu().numpy().astype(np.float64)
    want = et.psi_vjp(mo, flat, x.astype(np.float64), w1, 0.1 * w2)
    rel_ok(got, want, BOUND["L871"], "L871")
    if D == 2:
        # the matrix-core gradient path (forced onto this batch) carries the term too: the prior's reverse kernel adds (sum o) (b @ ob_to_b) to the three
        # channels of c and gives every raw output the adjoint of its channel's sum
        import os
        os.environ["WF_GRAD_TILE_MIN"] = "1"
        try:
            got_t = m.psi_vjp(x, w1, 0.1 * w2).cpu().numpy().astype(np.float64)
        finally:
            del os.environ["WF_GRAD_TILE_MIN"]
        assert not np.array_equal(got_t, got)
        assert rel_l2(got_t, want) < 5e-3 and rel_l2(got_t, got) < 5e-4, (rel_l2(got_t, want), rel_l2(got_t, got))
    got = m.logpdf_vjp(x, w1).cpu().numpy().astype(np.float64)
    want = et.logpdf_vjp(mo, flat, x.astype(np.float64), w1)
    assert rel_l2(got, want) < 2e-3, rel_l2(got, want)
    # the wave sampler evaluates the same head: direct(sample) is the latent point it drew (exact inverse), and the columns it draws follow
    # this psi^2 -- the one-lane-per-walker sampler (literal sequence) draws from the same distribution
    if D == 2:
        from scipy import stats
        xs, lat = m.sample(3, 4000, return_latent=True, exact=True)
        _, u = log_pdf(params, xs.cpu().numpy(), return_sample=True)
        d = np.abs(np.asarray(u) - lat.cpu().numpy()).max(1)
        assert np.median(d) < 1e-4 and np.quantile(d, 0.99) < 5e-3, (np.median(d), np.quantile(d, 0.99))
        import os
        os.environ["WF_WAVE_SAMPLE_MAX"] = "0"
        try:
            xs2, lat2 = m.sample(4, 4000, return_latent=True, exact=True)
        finally:
            del os.environ["WF_WAVE_SAMPLE_MAX"]
        for c in range(2):
            assert stats.ks_2samp(lat[:, c].cpu().numpy(), lat2[:, c].cpu().numpy()).pvalue > 1e-4
        # the staged large-batch sampler (forced onto this batch): the conditioner launch adds the term to the head's value channel
        os.environ["WF_SAMPLE_TILE_MIN"] = "1"
        try:
            xs3, lat3 = m.sample(5, 4000, return_latent=True, exact=True)
        finally:
            del os.environ["WF_SAMPLE_TILE_MIN"]
        _, u3 = log_pdf(params, xs3.cpu().numpy(), return_sample=True)
        d3 = np.abs(np.asarray(u3) - lat3.cpu().numpy()).max(1)
        assert np.median(d3) < 1e-4 and np.quantile(d3, 0.99) < 5e-3, (np.median(d3), np.quantile(d3, 0.99))
        for c in range(2):
            assert stats.ks_2samp(lat[:, c].cpu().numpy(), lat3[:, c].cpu().numpy()).pvalue > 1e-4


@pytest.mark.parametrize("D,knots", [(2, 23), (3, 23), (2, 33), (4, 23)])
def test_gated_wavefunction_energy_vs_autograd_oracle(D, knots):
    """wavefunctions.Waveflow with its own default set_nn_output_grad_to_zero=True (gated layers and prior): psi, H psi and the Laplacian
    from the wave forward sweep (the gate prod_{i<d} x_i^3 travels as a jet) against the torch oracle with the same four lines of
    model_factory.py:64-67; also through the R3 directional sweep (WF_ENERGY_R3).  33 knots: the 64-row layouts (two row blocks per
    dimension); D = 4: the MFMA kernel's staged mode."""
    import os
    import torch
    from oracle import energy_torch as et
    from waveflow_amd import flatten_params, flows, model_factory, wavefunctions
    mt = model_factory.get_masked_transform
    init = wavefunctions.Waveflow(
        flows.Serial(flows.BoxTransformLayer(3.0), *(flows.IMADE(mt(), 6, knots, 0.05, 1e-6, set_nn_output_grad_to_zero=True), flows.Reverse()) * 2),
        mt(allow_negative_params=True), 6, knots, constraints_dict_left={0: 0}, constraints_dict_right={0: 0},
        constrained_dimension_indices_left=list(range(D - 1)))
    params, psi, log_pdf, _ = init(6, D)
    flat = flatten_params(params)
    mo = et.TorchWaveflow(D, 2, "mean", 3.0, 6, knots, 0.05, tuple(range(D - 1)), dtype=torch.float64, i_gate=True, p_gate=True)
    x = sorted_walkers(96 if D < 4 else 48, D, 2.7, 17)
    m = psi.model
    m.ensure_params(params)
    pr = [0.0] * D
    ho, po, lo = et.hamiltonian(mo, flat, x.astype(np.float64), pr)
    for kernel in ("scalar", "wave", "mfma"):
        m.set_kernel(kernel)
        np.testing.assert_allclose(psi(params, x), po, rtol=0, atol=3e-5 * np.abs(po).max())
    m.set_kernel("auto")
    hp, ps, lap = m.hamiltonian(x, pr, return_psi=True, return_laplacian=True)
    np.testing.assert_allclose(ps, po, rtol=0, atol=3e-5 * np.abs(po).max())
    np.testing.assert_allclose(lap, lo, rtol=0, atol=3e-3 * np.abs(lo).max())
    np.testing.assert_allclose(hp, ho, rtol=0, atol=3e-3 * np.abs(ho).max())
    os.environ["WF_ENERGY_R3"] = "1"
    try:
        hp3, _, lap3 = m.hamiltonian(x, pr, return_psi=True, return_laplacian=True)
    finally:
        del os.environ["WF_ENERGY_R3"]
    np.testing.assert_allclose(lap3, lo, rtol=0, atol=3e-3 * np.abs(lo).max())
    # the ungated model with the same parameters has another Laplacian: the gate is not a no-op
    mo0 = et.TorchWaveflow(D, 2, "mean", 3.0, 6, knots, 0.05, tuple(range(D - 1)), dtype=torch.float64)
    assert np.abs(et.hamiltonian(mo0, flat, x.astype(np.float64), pr)[2] - lo).max() > 1e-2 * np.abs(lo).max()
    # ---- reverse sweep: the gate's adjoint reaches x (through prod x_i^3) and the zero_params leaves get their gradient
    g = np.random.default_rng(3)
    w1, w2 = g.normal(size=len(x)).astype(np.float32), g.normal(size=len(x)).astype(np.float32)
    is_zero = zero_leaf_mask(params)
    assert is_zero.sum() == (2 * (6 + knots) + (6 + knots - 1)) * D
    for wl in (np.zeros_like(w2), 0.1 * w2):          # psi only (first-order ring for the Laplacian weight 0 too), psi + Laplacian
        got = m.psi_vjp(x, w1, wl).cpu().numpy().astype(np.float64)
        want = et.psi_vjp(mo, flat, x.astype(np.float64), w1, wl)
        rel_ok(got, want, BOUND["L963"], "L963")
        assert rel_l2(got[is_zero], want[is_zero]) < 5e-3 and np.abs(want[is_zero]).max() > 0, rel_l2(got[is_zero], want[is_zero])
    got = m.logpdf_vjp(x, w1).cpu().numpy().astype(np.float64)
    want = et.logpdf_vjp(mo, flat, x.astype(np.float64), w1)
    assert rel_l2(got, want) < 2e-3 and rel_l2(got[is_zero], want[is_zero]) < 2e-3, (rel_l2(got, want), rel_l2(got[is_zero], want[is_zero]))
    # several chunks (a workspace for 40 walkers at a time) give the same gradient as one
    import ctypes
    from waveflow_amd import _lib
    L = _lib.lib()
    per = L.wf_psi_vjp_workspace_bytes(m._h, 1)
    xt = torch.as_tensor(x).cuda()
    wt, wlt = torch.as_tensor(w1).cuda(), torch.as_tensor(0.1 * w2).cuda()
    ws = torch.empty(per * 40, dtype=torch.uint8, device="cuda")
    gr = torch.empty(m.n_params, dtype=torch.float32, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(L.wf_psi_vjp(m._h, P(xt), len(x), P(wt), P(wlt), P(gr), P(ws), ws.numel(), None), "wf_psi_vjp")
    torch.cuda.synchronize()
    one = m.psi_vjp(x, w1, 0.1 * w2).cpu().numpy()
    assert rel_l2(gr.cpu().numpy().astype(np.float64), one.astype(np.float64)) < 1e-5
    # loss_fn_efficient's gradient (vqmc.py:193-221) of the gated model, and a few host-stepped training steps (the captured step is
    # not built for gated heads: the trainer falls back by itself)
    from waveflow_amd import vqmc
    from waveflow_amd.utils import physics
    sums, grad = m.vqmc_loss_grad(x, pr, running_average=-1.0)
    lo_, go_, _ = et.vqmc_loss_grad(mo, flat, x.astype(np.float64), pr, -1.0)
    assert rel_l2(grad.cpu().numpy().astype(np.float64), go_) < 1e-2
    # ---- the fused training step (sampler -> loss + gradient -> Adam -> image refill; wf_vqmc_train_step) on the gated model: two steps
    # replayed from one hipGraph equal the same sequence issued call by call (sampler with the step's stream, wf_vqmc_loss_grad, wf_adam_step)
    assert L.wf_vqmc_train_step_workspace_bytes(m._h, 128) > 0
    seed, Bt, lr = 77, 128, 1e-3

    def fresh():
        xs = torch.as_tensor(flat).cuda()
        return xs, torch.zeros_like(xs), torch.zeros_like(xs)

    xa, ma, va = fresh()
    st = m.make_train_state(xa, ma, va, 0, ring_len=8)
    m.set_params_device(xa)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.train_step(st, seed, Bt, pr, lr, exact_sampler=True)     # warm-up: workspace allocation
        xa.copy_(torch.as_tensor(flat)); ma.zero_(); va.zero_(); st["counter"].zero_(); st["ring"].zero_()
        m.set_params_device(xa)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            m.train_step(st, seed, Bt, pr, lr, exact_sampler=True)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay(); graph.replay()
    torch.cuda.synchronize()
    assert int(st["counter"].item()) == 2
    ring = st["ring"].cpu().numpy()
    xb, mb, vb = fresh()
    means = []
    for i in range(2):
        m.set_params_device(xb)
        # the step's sampler stream: seed advanced by the step counter (wf_kernels_wave.hip: seed += counter * 0x9E3779B97F4A7C15)
        xs_i = m.sample((seed + i * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF, Bt, exact=True)
        sums_i, grad_i = m.vqmc_loss_grad(xs_i, pr, running_average=0.0)
        m.adam_step(xb, grad_i, mb, vb, i, lr)
        means.append(float(sums_i[0] / sums_i[2]))
    torch.cuda.synchronize()
    # (Adam's bias corrections come from the host's powf in one path and the device's in the other: last-bit differences of the step size)
    assert np.abs(xa.cpu().numpy() - xb.cpu().numpy()).max() <= 2e-9 + 1e-6 * lr
    assert np.allclose([ring[0, 0] / ring[0, 2], ring[1, 0] / ring[1, 2]], means, rtol=1e-6)
    moved = np.abs(xa.cpu().numpy() - flat)
    assert moved[is_zero].max() > 1e-4 and moved[~is_zero].max() > 1e-4        # Adam moved the zero_params leaves too
    m.set_params(flat)


def zero_leaf_mask(params):
    """True at the entries of the flat vector that belong to a zero_params leaf (model_factory.py:84: the second leaf of a net's pair)."""
    from waveflow_amd import flatten_params
    import copy
    marked = copy.deepcopy(params)

    def mark(node, inside_zero):
        if isinstance(node, np.ndarray):
            node[...] = 1.0 if inside_zero else 0.0
            return
        if isinstance(node, (list, tuple)):
            is_pair = len(node) == 2 and isinstance(node[1], np.ndarray) and isinstance(node[0], (list, tuple))
            for i, c in enumerate(node):
                mark(c, inside_zero or (is_pair and i == 1))
    mark(marked, False)
    return flatten_params(marked) > 0.5


def test_trained_energy_respects_the_variational_bound(tmp_path):
    """Known answer for the derivative path (unpinned in the reference): the lowest antisymmetric eigenvalue of the 1-D
    soft-Coulomb He Hamiltonian in the box [-10, 10]^2 is -1.8161 (finite-difference diagonalisation, scratch/he1d_exact.py).
    The energy of the trained wavefunction on an independent |psi|^2 sample must lie just above it: a wrong Laplacian, potential
    or sampler shows up as a violation of the bound or as an energy far from it."""
    from waveflow_amd import vqmc
    t = vqmc.ModelTrainer(system_name="He", learning_rate=5e-4, box_length=10, num_epochs=20000, batch_size=512, log_every=10 ** 9)
    t.save_dir = str(tmp_path / "run")
    t.exact_sampler = True
    params, loss = t.start_training(verbose=False)
    m = t.psi.model
    m.ensure_params(params)
    es = []
    for seed in range(4):
        x = m.sample(500 + seed, 1 << 15, exact=True)
        h, ps = m.hamiltonian(x, t.h_fn.protons, return_psi=True)
        es.append((h / (ps + 1e-8)).double().cpu().numpy())
    e = np.concatenate(es)
    mean, sem = e.mean(), e.std() / np.sqrt(e.size)
    # the bound holds for the mean; closeness is judged by the median (the local energy has heavy tails near the nodes and the
    # box corners, a handful of walkers can move the mean of 1e5 samples by 0.05)
    assert mean > -1.8161 - 5 * sem, (mean, sem)
    assert -1.86 < np.median(e) < -1.75, (np.median(e), mean, sem)


def test_logpdf_loss_grad_matches_separate_calls():
    """wf_logpdf_loss_grad = log_pdf values + weight * logpdf_vjp(ones), from one sweep pair (several chunks too)."""
    import os
    import torch
    from conftest import GOLDEN
    from waveflow_amd import _lib, model_factory
    X = torch.as_tensor(np.load(os.path.join(GOLDEN, "circles_x256.npy")).astype(np.float32)).cuda()
    params, log_pdf, _ = model_factory.get_model(n_flow_layers=2, i_spline_reg=0.02, prior_constraint_dict_left={0: 0},
                                                 prior_constraint_dict_right={0: 0}, i_constraint_dict_left={0: 0.0},
                                                 i_constraint_dict_right={0: 1.0})(1, 2)
    m = log_pdf.model
    m.ensure_params(params)
    lp, grad = m.logpdf_loss_grad(X, -1.0 / 256)
    m.set_kernel("wave")
    assert torch.equal(lp, m.log_pdf(X))
    want = m.logpdf_vjp(X, torch.full((256,), -1.0 / 256, device="cuda"))
    assert torch.equal(grad, want)
    # a workspace for 64 rows only: four chunks, same values, gradient equal up to the grouping of the sums
    L = _lib.lib()
    per = L.wf_logpdf_vjp_workspace_bytes(m._h, 1)
    ws = torch.empty(per * 64, device="cuda", dtype=torch.uint8)
    lp2, g2 = torch.empty(256, device="cuda"), torch.empty(m.n_params, device="cuda")
    assert L.wf_logpdf_loss_grad(m._h, X.data_ptr(), 256, -1.0 / 256, lp2.data_ptr(), g2.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(lp2, lp) and rel_l2(g2.cpu().numpy(), grad.cpu().numpy()) < 1e-5


def test_mle_train_step_matches_separate_calls():
    """wf_mle_train_step = logpdf_loss_grad + adam_step + set_params_device (up to Adam's bias corrections being powf on the
    device and pow on the host); replayed from a hipGraph it is bitwise the eager sequence."""
    import os
    import torch
    from conftest import GOLDEN
    from waveflow_amd import model_factory
    from waveflow_amd.vqmc import flatten_params
    X = torch.as_tensor(np.load(os.path.join(GOLDEN, "circles_x256.npy")).astype(np.float32)).cuda()
    params, log_pdf, _ = model_factory.get_model(n_flow_layers=2, i_spline_reg=0.02, prior_constraint_dict_left={0: 0},
                                                 prior_constraint_dict_right={0: 0}, i_constraint_dict_left={0: 0.0},
                                                 i_constraint_dict_right={0: 1.0})(1, 2)
    m = log_pdf.model
    flat0 = torch.as_tensor(flatten_params(params).astype(np.float32)).cuda()
    n_steps, lr = 6, 1e-3

    # separate calls
    x, mm, vv = flat0.clone(), torch.zeros_like(flat0), torch.zeros_like(flat0)
    want_loss = []
    for step in range(1, n_steps + 1):
        m.set_params_device(x)
        lp, grad = m.logpdf_loss_grad(X, -1.0 / 256)
        s = m.block_sums(lp).cpu().tolist()
        want_loss.append(-s[0] / s[2])
        m.adam_step(x, grad, mm, vv, step, lr)

    def run(graphed):
        x2, m2, v2 = flat0.clone(), torch.zeros_like(flat0), torch.zeros_like(flat0)
        st = m.make_train_state(x2, m2, v2, 1, ring_len=4)
        m.set_params_device(x2)
        if graphed:
            m.mle_train_step(st, X, lr)     # sizes the workspace outside the capture
            x2.copy_(flat0); m2.zero_(); v2.zero_(); st["counter"].fill_(1)
            m.set_params_device(x2)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    m.mle_train_step(st, X, lr)
            torch.cuda.current_stream().wait_stream(side)
        got = []
        for step in range(1, n_steps + 1):
            graph.replay() if graphed else m.mle_train_step(st, X, lr)
            r = st["ring"].cpu().numpy()
            got.append(-r[step % 4, 0] / r[step % 4, 2])
        assert int(st["counter"].item()) == n_steps + 1
        return x2, m2, v2, got

    xe, me, ve, got_e = run(False)
    xg, mg, vg, got_g = run(True)
    assert torch.equal(xe, xg) and torch.equal(me, mg) and torch.equal(ve, vg) and got_e == got_g
    # (Adam turns a rounding difference in a near-zero gradient entry into a step-size difference: compare in bulk)
    dx = (xe - x).abs()
    assert float(torch.quantile(dx, 0.99)) < 1e-5 and float(dx.max()) < 2.5 * n_steps * lr, (float(torch.quantile(dx, 0.99)), float(dx.max()))
    np.testing.assert_allclose(got_e, want_loss, rtol=0, atol=1e-4)
    assert got_e[0] == want_loss[0] and want_loss[-1] < want_loss[0]
    x = xe
    # the images follow the state: the model evaluates with the trained parameters
    m.set_kernel("wave")
    lp_now = m.log_pdf(X)
    m.set_params_device(x)
    assert torch.equal(lp_now, m.log_pdf(X))


@pytest.mark.parametrize("D", [2, 3, 6])
def test_forward_laplacian_ring_gradient_matches_directional_ring(D, monkeypatch):
    """The taped sweeps run in RF (value, gradient, Laplacian / 2 per walker: D + 2 channels, one sample); WF_GRAD_R3 at model
    creation selects D samples of second-order Taylor coefficients instead.  Both are reverse mode over a Frobenius algebra
    (wf_ring.h): same local energies, same gradient up to rounding."""
    from waveflow_amd import model_factory
    x = sorted_walkers(300, D, 10.0, 9).astype(np.float32)
    protons = np.linspace(-3, 3, D)
    out = {}
    for tag in ("R3", "RF"):
        if tag == "R3":
            monkeypatch.setenv("WF_GRAD_R3", "1")
        else:
            monkeypatch.delenv("WF_GRAD_R3")
        init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                    n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=2, box_size=10.0)
        params, psi, log_pdf, sample = init_fun(5, D)
        m = psi.model
        m.ensure_params(params)
        sums, grad = m.vqmc_loss_grad(x, protons, -1.0)
        g = np.random.default_rng(2)
        wp, wl = g.normal(size=300).astype(np.float32), g.normal(size=300).astype(np.float32)
        out[tag] = (sums.cpu().numpy(), grad.cpu().numpy().astype(np.float64), m.psi_vjp(x, wp, wl).cpu().numpy().astype(np.float64))
    (s3, g3, v3), (sf, gf, vf) = out["R3"], out["RF"]
    np.testing.assert_allclose(sf, s3, rtol=2e-5)
    # (random signed weights cancel in psi_vjp: fp32 rounding of the two sweeps shows at the 1e-5 level)
    assert rel_l2(gf, g3) < 2e-5 and rel_l2(vf, v3) < 3e-4, (rel_l2(gf, g3), rel_l2(vf, v3))


def test_deferred_evaluation_tables_contract(he_flat):
    """wf_train_state.defer_eval_tables = 1: the steps keep the weight images current but leave the composite dimension-0 tables of
    the large-batch kernel to the next wf_model_set_params_device; the small-batch (wave) kernel, the sampler and further steps
    never see the difference."""
    import torch
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    protons = physics.system_catalogue[1]["He"][0].reshape(-1)
    x = torch.as_tensor(sorted_walkers(9000, 2, 9.0, 4)).cuda()
    out = {}
    for defer in (False, True):
        flat = torch.as_tensor(he_flat).cuda().clone()
        st = m.make_train_state(flat, torch.zeros_like(flat), torch.zeros_like(flat), 1, ring_len=8, defer_eval_tables=defer)
        m.set_params_device(flat)
        for _ in range(3):
            m.train_step(st, 99, 256, protons, 1e-3, exact_sampler=True)
        m.set_kernel("wave")
        small = m.psi(x[:2000])
        m.set_kernel("mfma")
        before = m.psi(x)
        # H psi of a large batch (>= 16384 walkers: the matrix-core tile path reads the MFMA image and the composite tables) straight
        # after the steps -- the header promises wf_hamiltonian_fwd needs no refresh: while the tables are stale it stays on the wave sweeps
        xe = torch.as_tensor(sorted_walkers(20000, 2, 9.0, 5)).cuda()
        h_before = torch.stack(m.hamiltonian(xe, protons, return_psi=True, return_laplacian=True))
        m.set_params_device(flat)
        after = m.psi(x)
        h_after = torch.stack(m.hamiltonian(xe, protons, return_psi=True, return_laplacian=True))
        m.set_kernel("auto")
        out[defer] = (flat.clone(), small, before, after, st["ring"].clone(), h_before, h_after)
    # same trajectory either way; the wave kernel and the refreshed large-batch kernel agree in both
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][4], out[True][4])
    assert torch.equal(out[False][1], out[True][1]) and torch.equal(out[False][3], out[True][3])
    assert torch.equal(out[False][2], out[False][3])            # not deferred: current on exit
    assert not torch.equal(out[True][2], out[True][3])          # deferred: stale until the refresh
    # H psi, psi, laplacian of 20000 walkers: identical trajectories -> the refreshed (tile-path) values are bit-equal in both runs, the
    # not-deferred run takes the tile path before and after, and the deferred run's wave-sweep values agree with the tile path's
    assert torch.equal(out[False][6], out[True][6]) and torch.equal(out[False][5], out[False][6])
    for k in range(3):
        a, b = out[True][5][k].double(), out[True][6][k].double()
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), (k, float((a - b).abs().max()), float(b.abs().max()))
    # (two fp32 kernels, each within 2.5e-5 of max|psi| of the reference's golden grid: their difference is bounded by the sum)
    np.testing.assert_allclose(out[True][3][:2000].cpu().numpy(), out[True][1].cpu().numpy(), rtol=0, atol=4e-5 * float(out[True][1].abs().max()))


@pytest.mark.gpu
def test_captured_large_batch_step_with_deferred_tables_replays_on_fresh_tables(he_flat):
    """a hipGraph of wf_vqmc_train_step captured with defer_eval_tables = 1 at a batch size of the matrix-core sampler /
    gradient (>= 16 384 walkers) bakes k_tsample / k_efused / k_ebwd in; a deferred refresh would leave every replay after the first on the
    MFMA image and the composite tables of older parameters.  The step now refreshes everything whenever such a path applies at its batch
    size, whatever the flag says: three replays of the captured deferred step equal three eager steps without deferral, bit for bit."""
    import torch
    from waveflow_amd.utils import physics
    params, psi, log_pdf, sample = he(he_flat)
    m = psi.model
    protons = physics.system_catalogue[1]["He"][0].reshape(-1)
    B, out = 16384, {}
    for mode in ("eager", "graph"):
        flat = torch.as_tensor(he_flat).cuda().clone()
        st = m.make_train_state(flat, torch.zeros_like(flat), torch.zeros_like(flat), 1, ring_len=8, defer_eval_tables=(mode == "graph"))
        m.set_params_device(flat)
        m.train_step(st, 5, B, protons, 1e-3, exact_sampler=True)       # (also sizes the workspace and the scratch outside the capture)
        if mode == "eager":
            for _ in range(3):
                m.train_step(st, 5, B, protons, 1e-3, exact_sampler=True)
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    m.train_step(st, 5, B, protons, 1e-3, exact_sampler=True)
            torch.cuda.current_stream().wait_stream(side)
            # (the capture itself executes nothing: three replays = steps 2 .. 4)
            for _ in range(3):
                graph.replay()
        torch.cuda.synchronize()
        out[mode] = (flat.clone(), st["ring"].clone())
    assert torch.isfinite(out["graph"][0]).all()
    assert torch.equal(out["eager"][0], out["graph"][0]) and torch.equal(out["eager"][1], out["graph"][1])
