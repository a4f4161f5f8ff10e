"""The local-energy oracle (oracle/energy_torch.py): value path vs the pinned C oracle; derivative machinery on a case
with a known answer (CPU)."""
import numpy as np
import torch

import oracle
from oracle import energy_torch as et


def test_torch_psi_equals_c_oracle(golden, he_flat):
    sp = np.sort(golden["he_golden"]["sample_points"], -1).astype(np.float64)
    m = et.he_model()
    ps = m.psi(he_flat, torch.tensor(sp)).numpy()
    ref = oracle.he_model(10.0).psi(he_flat, sp, f64=True)
    assert np.abs(ps - ref).max() < 5e-7        # fp64 arithmetic vs the f64 oracle build rounded to fp32


def test_custom_derivative_rule_is_what_autograd_sees():
    """d/dx of the order-nd lerp must be the order-(nd+1) lerp (isplines_jax.py:60-66), to second order."""
    I = torch.tensor(np.asarray(oracle.table(oracle.KIND_I, 5, 16), dtype=np.float32)).double()
    x = torch.tensor([0.1234, 0.5, 0.77], dtype=torch.float64, requires_grad=True)
    c = torch.rand(3, I.shape[1], dtype=torch.float64)
    y = et.table_spline(x, c, I, 0)
    (g,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    np.testing.assert_allclose(g.detach(), et.table_spline(x, c, I, 1).detach(), rtol=1e-12)
    (g2,) = torch.autograd.grad(g.sum(), x)
    np.testing.assert_allclose(g2, et.table_spline(x, c, I, 2).detach(), rtol=1e-12)


def test_potential_matches_formula():
    x = torch.tensor([[0.3, -1.2], [2.0, 2.5]], dtype=torch.float64)
    v = et.potential(x, torch.tensor([0.0, 0.0], dtype=torch.float64)).numpy()
    ref = [-2 / np.sqrt(1 + 0.09) - 2 / np.sqrt(1 + 1.44) + 1 / np.sqrt(1 + 2.25),
           -2 / np.sqrt(5.0) - 2 / np.sqrt(7.25) + 1 / np.sqrt(1.25)]
    np.testing.assert_allclose(v, ref, rtol=1e-12)


def test_he_local_energy_scale_matches_the_shipped_run(golden, he_flat):
    """Soft pin: the shipped loss trace (mean local energy per epoch) ends at -441 +- several hundred (a diverged run,
    BASELINE.md); the local energies of the shipped samples under the shipped checkpoint must be of that size."""
    sp = np.sort(golden["he_golden"]["sample_points"], -1).astype(np.float64)
    h, p, lap = et.hamiltonian(et.he_model(), he_flat, sp, [0.0, 0.0])
    el = h / (p + 1e-8)
    assert -1500 < np.median(el) < -100 and np.isfinite(el).all()
