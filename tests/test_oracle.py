"""The oracle against the reference's own fixtures (CPU)."""
import numpy as np
import pytest

import oracle
from conftest import he_grid


def test_tables_bit_exact_vs_reference_fixtures(golden):
    g = golden["ref_tables_k5_n16"]
    I = oracle.table(oracle.KIND_I, 5, 16)
    B = oracle.table(oracle.KIND_B, 5, 16)
    for nd in range(4):
        assert np.array_equal(I[nd], g[f"I_nd{nd}"])
        assert np.array_equal(B[nd], g[f"B_nd{nd}"])


def test_tables_bit_exact_vs_reference_probes(golden):
    p = golden["ref_probes"]
    cols = p["cols"]
    n = 0
    for key in p.files:
        if key[0] in "IBM" and key[1] == "_":
            kind = {"M": 0, "I": 1, "B": 2}[key[0]]
            k, kn = int(key.split("_")[1][1:]), int(key.split("_")[2][1:])
            assert np.array_equal(oracle.table(kind, k, kn)[:, :, cols], p[key]), key
            n += 1
    assert n >= 10


def test_gram_schmidt_restatement_vs_reference(golden):
    p = golden["ref_probes"]
    np.testing.assert_allclose(oracle.gram_schmidt_symm(p["gs_small_in"]), p["gs_small_out"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(oracle.gram_schmidt_l2r(p["gs_small_in"]), p["gs_l2r_small_out"], rtol=0, atol=1e-13)
    B = oracle.table(oracle.KIND_B, 6, 23)[0]
    ob = oracle.gram_schmidt_symm(B.T).T
    np.testing.assert_allclose(ob[:, p["gs_B_k6_n23_cols"]], p["gs_B_k6_n23_out_sub"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(ob @ ob.T, p["gs_B_k6_n23_out_gram"], rtol=0, atol=1e-9)


def test_ortho_tables_vs_reference_fixtures(golden):
    g = golden["ref_tables_k5_n16"]
    Bt, OB, b2o, o2b = oracle.ortho_b(5, 16)
    np.testing.assert_allclose(b2o, g["b_to_ob"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(o2b, g["ob_to_b"], rtol=0, atol=1e-13)
    cols = g["OB_cols"]
    for nd in range(4):
        ref = g[f"OB_nd{nd}_sub"]
        np.testing.assert_allclose(OB[nd][:, cols], ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    # identities the survey verified on the fixtures
    np.testing.assert_allclose(o2b @ b2o, np.eye(20), atol=1e-12)
    np.testing.assert_allclose(OB[0] @ OB[0].T / 2000, np.eye(20), atol=1e-12)


def test_he_checkpoint_psi_vs_reference_outputs(golden, he_flat):
    """The end-to-end pin: psi of the shipped He checkpoint on the grids the reference saved (helpers.py:52-84)."""
    g = golden["he_golden"]
    m = oracle.he_model(10.0)
    assert m.n_params() == he_flat.size == 32588
    coords, srt, sign = he_grid()
    psi = m.psi(he_flat, srt) * sign
    err = np.abs(psi - g["psi_grid"])
    assert err.max() < 2e-5, err.max()          # max|psi| = 1.53
    assert np.median(err) < 1e-7
    for nm in ("onproton", "random"):
        c = g[nm + "_coord"]
        s = (-1.0) ** (c[:, 0] > c[:, 1])
        ps = m.psi(he_flat, np.sort(c, -1)) * s
        assert np.abs(ps - g[nm + "_values"]).max() < 1e-5


def test_he_logpdf_consistent_with_psi(golden, he_flat):
    m = oracle.he_model(10.0)
    sp = np.sort(golden["he_golden"]["sample_points"], -1)
    lp, u = m.log_pdf(he_flat, sp, return_u=True)
    ps = m.psi(he_flat, sp)
    assert u.min() > 0 and u.max() < 1
    np.testing.assert_allclose(np.log(ps.astype(np.float64) ** 2), lp, atol=2e-5)
    # normalisation: the reference only prints it (tests/test_waveflow.py:52)
    L, n = 10.0, 200
    xs = np.linspace(-L, L, n)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    c = np.stack([X, Y], -1).reshape(-1, 2)
    p = m.psi(he_flat, np.sort(c, -1)).astype(np.float64)
    assert abs((p ** 2).sum() * (2 * L / (n - 1)) ** 2 - 1.0) < 0.03


def test_bin_index_definition():
    """x_l = floor(x*(n_mesh-1)), x_r = ceil(x*(n_mesh-1)) in fp32 (isplines_jax.py:46-48)."""
    m = oracle.Model(D=2, n_layers=1, i_k=5, i_knots=16, prior="uniform")
    p = m.init_params(0)
    u = np.random.default_rng(0).uniform(0, 1, size=(1000, 2)).astype(np.float32)
    u[:4] = [[0.0, 1.0], [0.5, 0.25], [1.0 / 1999, 1998.0 / 1999], [np.float32(1e-7), np.float32(1 - 1e-7)]]
    _, _, idx = m.imade_direct(p, u)
    xs = u * np.float32(1999)
    assert np.array_equal(idx[..., 0], np.floor(xs).astype(np.int32))
    assert np.array_equal(idx[..., 1], np.ceil(xs).astype(np.int32))


def test_imade_is_monotone_bijection_of_unit_interval():
    m = oracle.Model(D=2, n_layers=1, i_k=6, i_knots=23, i_reg=0.05, i_left={0: 0}, i_right={0: 1}, prior="uniform")
    p = m.init_params(3)
    t = np.linspace(0, 1, 400, dtype=np.float32)
    u = np.stack([np.full_like(t, 0.3), t], -1)
    y, ld, _ = m.imade_direct(p, u)
    assert abs(y[0, 1]) < 1e-6 and abs(y[-1, 1] - 1) < 1e-5
    assert np.all(np.diff(y[:, 1]) > -1e-6)
    # log-det vs finite differences of the table-lerp map (dim 1 only varies)
    fd = np.diff(y[:, 1].astype(np.float64)) / np.diff(t.astype(np.float64))
    mid = 0.5 * (np.exp(ld[1:].astype(np.float64)) + np.exp(ld[:-1].astype(np.float64)))
    # ld sums both dims; dim 0 is constant along the scan
    ratio = fd / mid
    assert np.std(ratio[5:-5]) / np.mean(ratio[5:-5]) < 0.02


def test_rqs_self_consistency():
    g = np.random.default_rng(0)
    K = 8
    for _ in range(50):
        uw, uh, ud = g.normal(size=K), g.normal(size=K), g.normal(size=K + 1)
        x = float(g.uniform(0.01, 0.99))
        y, ld, b = oracle.rqs(x, uw, uh, ud)
        x2, ld2, b2 = oracle.rqs(y, uw, uh, ud, inverse=True)
        assert abs(x2 - x) < 2e-5 + 1e-6 / np.exp(ld) and abs(ld + ld2) < 1e-3 and b == b2
        eps = 1e-3
        y1, _, _ = oracle.rqs(x + eps, uw, uh, ud)
        y0, _, _ = oracle.rqs(x - eps, uw, uh, ud)
        assert abs(np.log((y1 - y0) / (2 * eps)) - ld) < 5e-2
