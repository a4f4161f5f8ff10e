"""C-ABI surface and host logic (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
from waveflow_amd import _lib, checkpoint, core, flows, model_factory
from conftest import ROOT


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "waveflow_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(wf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.EXPORTS)
    assert _lib.lib().wf_abi_version() == 2     # round 2: gate flags and the coupling-stack fields appended to wf_model_desc


def test_desc_struct_matches_header_layout():
    # 4-byte fields only; guards against drift between _lib.ModelDesc and wf_model_desc
    assert ctypes.sizeof(_lib.BC) == 4 + 4 * 4 + 4 * 4
    assert ctypes.sizeof(_lib.ModelDesc) == 4 * 9 + 2 * 36 + 4 * 3 + 2 * 36 + 4 + 4 + 16 * 4 + 4 + 4 + 2 * 4 + 4 * 4   # ... i_gate, p_gate; nsc_*
    assert _lib.ModelDesc.i_gate.offset == _lib.ModelDesc.i_reverse_tol.offset + 4 and _lib.ModelDesc.nsc_reverse.offset == ctypes.sizeof(_lib.ModelDesc) - 4
    # wf_train_state: six pointers + int32 (padded to 8)
    assert ctypes.sizeof(_lib.TrainState) == 6 * 8 + 8 and _lib.TrainState.ring_len.offset == 48 and _lib.TrainState.defer_eval_tables.offset == 52


def test_strerror_and_no_device_is_loud():
    L = _lib.lib()
    assert b"no gfx950" in L.wf_strerror(-4)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d = _lib.ModelDesc()
    h = ctypes.c_void_p()
    assert L.wf_model_create(ctypes.byref(d), 0, ctypes.byref(h)) == _lib.ERR_NO_DEVICE
    init_fun = model_factory.get_waveflow_model(2)
    with pytest.raises(_lib.WfError):
        init_fun(0, 2)


def test_param_tree_layout_matches_reference_checkpoint(he_flat):
    """Leaf order / shapes of the He checkpoint: 3 x [(W0,b0),(W1,b1),(W2,b2),zero] + prior."""
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, n_flow_layers=3, box_size=10)
    g = flows.as_generator(0)
    tparams = init_fun.transformation.init_params(g, 2)
    sparams = init_fun.sp.init_params(g, 2, 23 + 6 - 1)
    tree = (tparams, sparams)
    shapes = [tuple(a.shape) for a in core.tree_leaves(tree)]
    per_layer = [(2, 64), (64,), (64, 64), (64,), (64, 58), (58,), (2, 29)]
    assert shapes == per_layer * 3 + [(2, 64), (64,), (64, 64), (64,), (64, 56), (56,), (2, 28)]
    assert core.flatten_params(tree).size == he_flat.size == 32588
    assert tparams[0] == () and tparams[2] == ()  # Box and Reverse carry no params
    rebuilt = checkpoint.unflatten_like(tree, he_flat)
    assert np.array_equal(core.flatten_params(rebuilt), he_flat)
    assert oracle.he_model().n_params() == he_flat.size


def test_serial_parsing():
    mt = model_factory.get_masked_transform()
    s = flows.Serial(flows.BoxTransformLayer(3.0), flows.IMADE(mt, 5, 16), flows.Reverse(), flows.IMADE(mt, 5, 16), flows.Reverse())
    box, layers = flows.parse_serial(s.spec)
    assert box.box_side == 3.0 and len(layers) == 2
    d = flows._desc(2, layers, box=box)
    assert (d.n_flow_layers, d.box_kind, d.i_degree, d.i_knots, d.i_right.n, d.i_right.value[0]) == (2, 1, 5, 16, 1, 1.0)
    assert flows.parse_serial(flows.Serial(flows.Reverse(), flows.IMADE(mt, 5, 16)).spec) is None
    with pytest.raises(NotImplementedError):
        flows.IMADE(model_factory.get_masked_transform(allow_negative_params=True))


def test_inversion_count():
    from waveflow_amd.utils.coordinates import get_num_inversion_count
    c = np.array([[1.0, 2.5, 2.0, -3.0], [0.0, -1.5, 2.0, -3.0], [0.0, 1.0, 2.0, 3.0]])
    assert get_num_inversion_count(c).tolist() == [4, 4, 0]


def test_round2_host_logic_specs_and_optimizer_state(tmp_path):
    """Host-side pieces added in round 2 (no GPU): coupling-stack parsing, the one-mesh rule, Adam moments next to the checkpoints."""
    from waveflow_amd import vqmc
    from waveflow_amd.flows import IMADESpec, NSCSpec, ReverseSpec, SerialSpec, parse_nsc_serial, parse_serial, _desc
    a = NSCSpec(5, 3.0, 8)
    assert parse_nsc_serial(SerialSpec([a, ReverseSpec(), a, ReverseSpec()])) == (a, 2, True)
    assert parse_nsc_serial(SerialSpec([a, a, a])) == (a, 3, False)
    assert parse_nsc_serial(SerialSpec([a, ReverseSpec(), a])) is None                    # ragged
    assert parse_nsc_serial(SerialSpec([a, NSCSpec(8, 3.0, 8)])) is None                   # layers differ
    assert parse_nsc_serial(SerialSpec([IMADESpec(5, 15, 0.0, 1e-6, {}, {}), ReverseSpec()])) is None
    assert parse_serial(SerialSpec([a, ReverseSpec()])) is None                            # not a conditioner-net stack either way
    # one mesh size per fused model: a prior spline on another mesh than the layers' is refused, not silently re-meshed
    lay = IMADESpec(5, 15, 0.0, 1e-6, {0: 0.0}, {0: 1.0}, 1000)
    with pytest.raises(NotImplementedError):
        _desc(2, [lay], prior=_lib.PRIOR_MFLOW, p_degree=5, p_knots=15, n_mesh=2000)
    d = _desc(2, [lay], prior=_lib.PRIOR_MFLOW, p_degree=5, p_knots=15, n_mesh=1000, p_gate=True)
    assert d.n_mesh == 1000 and d.p_gate == 1 and d.i_gate == 0
    assert _desc(2, [IMADESpec(5, 15, 0.0, 1e-6, {}, {}, 2000, True)]).i_gate == 1
    # optimiser state: saved per checkpoint epoch, restored only for exactly that epoch and shape
    st = vqmc.OptState(None, np.zeros(7, np.float32), np.arange(7, dtype=np.float32), np.arange(7, dtype=np.float32) ** 2)
    vqmc._save_optimizer_state(str(tmp_path), st, 300)
    fresh = vqmc.OptState(None, np.zeros(7, np.float32), np.zeros(7, np.float32), np.zeros(7, np.float32))
    assert not vqmc._load_optimizer_state(str(tmp_path), fresh, 299) and fresh.m.sum() == 0
    assert vqmc._load_optimizer_state(str(tmp_path), fresh, 300)
    assert np.array_equal(fresh.m, st.m) and np.array_equal(fresh.v, st.v)
    other = vqmc.OptState(None, np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(8, np.float32))
    assert not vqmc._load_optimizer_state(str(tmp_path), other, 300)
    assert not vqmc._load_optimizer_state(str(tmp_path / "nowhere"), fresh, 300)


def test_table_cache_files_are_the_reference_fixture_files(tmp_path, golden):
    """utils/table_cache.py writes the reference's cache (isplines_jax.py:115, bsplines_jax.py:76-80 naming); for (k = 5, 16 internal knots,
    2000 mesh points) the reference ships its own files (waveflow/tests/splines/cached_bases/{I,B}, packed in tests/golden): same names,
    same shapes, I and plain-B values bit-equal, orthogonal tables and change-of-basis matrices to 1e-10."""
    from waveflow_amd.utils import table_cache as tc
    g = golden["ref_tables_k5_n16"]
    assert tc.cache_file_names("I", 5, 16)["nd"][1] == "degree_5_niknots_21_nmp_2000_nd_1.npy"
    nb = tc.cache_file_names("B", 5, 16)
    assert nb["nd"][0] == "b_degree_5_niknots_21_nmp_2000_nd_0.npy" and nb["ob"][3] == "ob_degree_5_niknots_21_nmp_2000_nd_3.npy"
    assert nb["b_to_ob"] == "degree_5_niknots_21_nmp_2000_b_to_ob.npy" and nb["ob_to_b"] == "degree_5_niknots_21_nmp_2000_ob_to_b.npy"
    wi = tc.write_cached_bases(str(tmp_path / "I"), "I", 5, 16)
    wb = tc.write_cached_bases(str(tmp_path / "B"), "B", 5, 16)
    assert len(wi) == 4 and len(wb) == 10
    I = tc.load_cached_bases(str(tmp_path / "I"), "I", 5, 16)
    B, OB, b2o, o2b = tc.load_cached_bases(str(tmp_path / "B"), "B", 5, 16)
    for nd in range(4):
        assert I[nd].dtype == np.float64 and I[nd].shape == g[f"I_nd{nd}"].shape and np.array_equal(I[nd], g[f"I_nd{nd}"])
        assert B[nd].shape == g[f"B_nd{nd}"].shape and np.array_equal(B[nd], g[f"B_nd{nd}"])
        sub = OB[nd][:, g["OB_cols"]]
        assert np.abs(sub - g[f"OB_nd{nd}_sub"]).max() <= 1e-10 * max(1.0, np.abs(g[f"OB_nd{nd}_sub"]).max())
    assert np.abs(b2o - g["b_to_ob"]).max() <= 1e-10 * np.abs(g["b_to_ob"]).max()
    assert np.abs(o2b - g["ob_to_b"]).max() <= 1e-10 * np.abs(g["ob_to_b"]).max()
    m = tc.write_cached_bases(str(tmp_path / "M"), "M", 3, 15)
    assert os.path.basename(m[0]) == "degree_3_niknots_16_nmp_2000_nd_0.npy" and np.load(m[0]).shape == (16, 2000)   # n_knots - k = 15 + 2 (k - 1) - k
    with pytest.raises(FileNotFoundError):
        tc.load_cached_bases(str(tmp_path / "I"), "I", 6, 23)
