"""Product table builder (wf_tables_build, host-only entry point of the C ABI) against the reference fixtures."""
import numpy as np

import oracle
from waveflow_amd import _lib, build_tables


def test_raw_tables_bit_exact_vs_reference_fixtures(golden):
    g = golden["ref_tables_k5_n16"]
    I = build_tables(_lib.SPLINE_I, 5, 16)
    B = build_tables(_lib.SPLINE_B, 5, 16)
    assert I.shape == (4, 21, 2000) and B.shape == (4, 20, 2000)
    for nd in range(4):
        assert np.array_equal(I[nd], g[f"I_nd{nd}"])
        assert np.array_equal(B[nd], g[f"B_nd{nd}"])


def test_raw_tables_bit_exact_vs_reference_probes(golden):
    p = golden["ref_probes"]
    cols = p["cols"]
    for key in p.files:
        if key[0] in "IBM" and key[1] == "_":
            kind = {"M": _lib.SPLINE_M, "I": _lib.SPLINE_I, "B": _lib.SPLINE_B}[key[0]]
            k, kn = int(key.split("_")[1][1:]), int(key.split("_")[2][1:])
            assert np.array_equal(build_tables(kind, k, kn)[:, :, cols], p[key]), key


def test_raw_tables_equal_oracle_for_other_shapes():
    for kind in (0, 1, 2):
        for k, n, nm in ((2, 5, 200), (4, 7, 333), (8, 12, 500), (6, 33, 2000)):
            assert np.array_equal(build_tables(kind, k, n, nm), oracle.table(kind, k, n, nm)), (kind, k, n, nm)


def test_ortho_tables_vs_reference_fixtures(golden):
    g = golden["ref_tables_k5_n16"]
    OB, b2o, o2b = build_tables(_lib.SPLINE_OB, 5, 16)
    np.testing.assert_allclose(b2o, g["b_to_ob"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(o2b, g["ob_to_b"], rtol=0, atol=1e-13)
    cols = g["OB_cols"]
    for nd in range(4):
        ref = g[f"OB_nd{nd}_sub"]
        np.testing.assert_allclose(OB[nd][:, cols], ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def test_ortho_tables_he_shape_vs_oracle():
    OB, b2o, o2b = build_tables(_lib.SPLINE_OB, 6, 23)
    _, OBo, b2oo, o2bo = oracle.ortho_b(6, 23)
    np.testing.assert_allclose(OB[0], OBo[0], rtol=0, atol=1e-12 * np.abs(OBo[0]).max())
    np.testing.assert_allclose(o2b, o2bo, rtol=0, atol=1e-13)
    np.testing.assert_allclose(OB[0] @ OB[0].T / 2000, np.eye(28), atol=1e-11)


def test_table_errors():
    L = _lib.lib()
    assert L.wf_tables_build(7, 5, 16, 2000, None, None, None) == -1
    assert L.wf_tables_build(_lib.SPLINE_I, 0, 16, 2000, None, None, None) == -1
    # odd number of B bases: the reference exits (ortho_splines.py:58-63)
    nb = L.wf_tables_build(_lib.SPLINE_OB, 5, 15, 2000, None, None, None)
    assert nb == 19
    out = np.zeros((4, nb, 2000))
    assert L.wf_tables_build(_lib.SPLINE_OB, 5, 15, 2000, out.ctypes.data, None, None) == -6
