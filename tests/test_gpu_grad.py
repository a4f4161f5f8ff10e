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
    assert rel_l2(grad.cpu().numpy().astype(np.float64), go) < 5e-3


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
    # a workspace that only holds 128 samples forces 16 chunks: same gradient up to the order of the atomic sums
    per = L.wf_psi_vjp_workspace_bytes(m._h, 1) // 64
    ws = torch.empty(per * 128, device="cuda", dtype=torch.uint8)
    grad = torch.empty(m.n_params, device="cuda")
    w2 = (w * 0.1).contiguous()
    rc = L.wf_psi_vjp(m._h, x.data_ptr(), 1000, w.data_ptr(), w2.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_l2(grad.cpu().numpy(), full.cpu().numpy()) < 1e-4
    assert L.wf_psi_vjp(m._h, x.data_ptr(), 1000, w.data_ptr(), w2.data_ptr(), grad.data_ptr(), ws.data_ptr(), 16, None) == -1
    assert m.psi_vjp(np.zeros((0, 2), np.float32), np.zeros(0), np.zeros(0)).abs().sum().item() == 0.0
