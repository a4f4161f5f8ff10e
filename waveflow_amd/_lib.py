"""ctypes binding of libwaveflow_hip.so (include/waveflow_hip.h).  No fallback: if the library
is missing or a call fails, an exception is raised."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libwaveflow_hip.so")
# WF_LIB: experiment builds under scratch/variants only (A/B timing scripts).  It is honoured only together with WF_LIB_EXPERIMENT=1 and the
# loaded path is reported on stderr: tests/conftest.py builds and validates the default library, nothing else.
if os.environ.get("WF_LIB") and os.environ.get("WF_LIB_EXPERIMENT") != "1":
    raise ImportError("WF_LIB is set without WF_LIB_EXPERIMENT=1: refusing to load a library other than " + DEFAULT_LIB)
LIB_PATH = os.environ.get("WF_LIB") or DEFAULT_LIB

WF_MAX_DIM, WF_MAX_BC = 16, 4
SPLINE_M, SPLINE_I, SPLINE_B, SPLINE_OB = 0, 1, 2, 3
LAYER_IMADE, LAYER_MADE, LAYER_NSC = 0, 1, 2
BOX_NONE, BOX_MEAN, BOX_FIRST = 0, 1, 2
PRIOR_WAVEFLOW, PRIOR_MFLOW, PRIOR_UNIFORM, PRIOR_NORMAL = 0, 1, 2, 3
KERNEL_AUTO, KERNEL_SCALAR, KERNEL_MFMA, KERNEL_WAVE = 0, 1, 2, 3
ERR_NO_DEVICE = -4
ERR_UNSUPPORTED = -2


class WfError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        L = lib()
        msg = L.wf_strerror(status).decode()
        if status == -3:
            msg += ": " + L.wf_last_hip_error_string().decode()
        super().__init__(f"{what}: {msg} (status {status})")


class BC(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("n_derivative", ctypes.c_int32 * WF_MAX_BC), ("value", ctypes.c_float * WF_MAX_BC)]

    @classmethod
    def from_dict(cls, d):
        bc = cls()
        d = {} if d is None else d
        if len(d) > WF_MAX_BC:
            raise ValueError("at most %d boundary constraints per side" % WF_MAX_BC)
        bc.n = len(d)
        for i, (nd, v) in enumerate(d.items()):
            bc.n_derivative[i] = int(nd)
            bc.value[i] = float(v)
        return bc


class ModelDesc(ctypes.Structure):
    _fields_ = [("n_dim", ctypes.c_int32), ("hidden", ctypes.c_int32), ("n_flow_layers", ctypes.c_int32),
                ("layer_kind", ctypes.c_int32), ("box_kind", ctypes.c_int32), ("box_size", ctypes.c_float),
                ("i_degree", ctypes.c_int32), ("i_knots", ctypes.c_int32), ("i_reg", ctypes.c_float),
                ("i_left", BC), ("i_right", BC), ("prior_kind", ctypes.c_int32), ("p_degree", ctypes.c_int32),
                ("p_knots", ctypes.c_int32), ("p_left", BC), ("p_right", BC), ("normal_offset", ctypes.c_float),
                ("n_constrained_left", ctypes.c_int32), ("constrained_left", ctypes.c_int32 * WF_MAX_DIM),
                ("n_mesh", ctypes.c_int32), ("i_reverse_tol", ctypes.c_float), ("i_gate", ctypes.c_int32), ("p_gate", ctypes.c_int32),
                ("nsc_bins", ctypes.c_int32), ("nsc_tail_bound", ctypes.c_float), ("nsc_hidden", ctypes.c_int32), ("nsc_reverse", ctypes.c_int32)]


class TrainState(ctypes.Structure):
    """wf_train_state (include/waveflow_hip.h)"""
    _fields_ = [("params_dev", ctypes.c_void_p), ("m_dev", ctypes.c_void_p), ("v_dev", ctypes.c_void_p), ("counter_dev", ctypes.c_void_p),
                ("running_average_dev", ctypes.c_void_p), ("loss_ring_dev", ctypes.c_void_p), ("ring_len", ctypes.c_int32), ("defer_eval_tables", ctypes.c_int32)]


EXPORTS = ["wf_abi_version", "wf_strerror", "wf_last_hip_error", "wf_last_hip_error_string", "wf_device_count",
           "wf_tables_build", "wf_model_create", "wf_model_destroy", "wf_model_param_count", "wf_model_n_bases",
           "wf_model_set_params", "wf_model_set_kernel", "wf_logpdf_fwd", "wf_psi_fwd", "wf_flow_fwd", "wf_layer_fwd",
           "wf_block_sums", "wf_block_sums_workspace_bytes", "wf_rqs_fwd", "wf_inverse_fwd", "wf_sample", "wf_hamiltonian_fwd",
           "wf_psi_vjp", "wf_psi_vjp_workspace_bytes", "wf_vqmc_seeds",
           "wf_logpdf_vjp", "wf_logpdf_vjp_workspace_bytes", "wf_vqmc_loss_grad", "wf_model_set_params_device", "wf_adam_step",
           "wf_vqmc_train_step", "wf_vqmc_train_step_workspace_bytes", "wf_nsc_fwd", "wf_nsc_workspace_bytes", "wf_logpdf_loss_grad", "wf_mle_train_step", "wf_mle_train_step_workspace_bytes", "wf_vqmc_train_step_local",
           "wf_vqmc_train_step_apply", "wf_psi_antisym_fwd", "wf_logpdf_unsorted_fwd", "wf_inversion_count"]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built; run `python -m waveflow_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback.")
    # PyTorch ships its own libamdhip64.so.7; importing torch first makes this library bind to that same
    # HIP runtime (one runtime per process: streams and device pointers are shared with torch).
    import torch  # noqa: F401
    if LIB_PATH != DEFAULT_LIB:
        import sys
        print(f"waveflow_amd: experiment library {LIB_PATH}", file=sys.stderr)
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32p = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    L.wf_abi_version.restype = i32
    L.wf_strerror.restype = ctypes.c_char_p
    L.wf_strerror.argtypes = [i32]
    L.wf_last_hip_error.restype = i32
    L.wf_last_hip_error_string.restype = ctypes.c_char_p
    L.wf_device_count.restype = i32
    L.wf_tables_build.restype = i32
    L.wf_tables_build.argtypes = [i32, i32, i32, i32, vp, vp, vp]
    L.wf_model_create.restype = i32
    L.wf_model_create.argtypes = [ctypes.POINTER(ModelDesc), i32, ctypes.POINTER(vp)]
    L.wf_model_destroy.restype = None
    L.wf_model_destroy.argtypes = [vp]
    L.wf_model_param_count.restype = i64
    L.wf_model_param_count.argtypes = [vp]
    L.wf_model_n_bases.restype = i32
    L.wf_model_n_bases.argtypes = [vp, i32]
    L.wf_model_set_params.restype = i32
    L.wf_model_set_params.argtypes = [vp, f32p, i64, vp]
    L.wf_model_set_kernel.restype = i32
    L.wf_model_set_kernel.argtypes = [vp, i32]
    for name in ("wf_logpdf_fwd", "wf_psi_fwd"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.wf_flow_fwd.restype = i32
    L.wf_flow_fwd.argtypes = [vp, vp, i64, vp, vp, vp]
    L.wf_psi_antisym_fwd.restype = i32
    L.wf_psi_antisym_fwd.argtypes = [vp, vp, i64, vp, vp, vp]
    L.wf_logpdf_unsorted_fwd.restype = i32
    L.wf_logpdf_unsorted_fwd.argtypes = [vp, vp, i64, vp, vp]
    L.wf_inversion_count.restype = i32
    L.wf_inversion_count.argtypes = [vp, i64, i32, vp, vp]
    L.wf_layer_fwd.restype = i32
    L.wf_layer_fwd.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp]
    L.wf_inverse_fwd.restype = i32
    L.wf_inverse_fwd.argtypes = [vp, vp, i64, vp, i32, vp]
    L.wf_sample.restype = i32
    L.wf_sample.argtypes = [vp, ctypes.c_uint64, i64, vp, vp, i32, vp]
    L.wf_hamiltonian_fwd.restype = i32
    L.wf_hamiltonian_fwd.argtypes = [vp, vp, i64, vp, i32, vp, vp, vp, vp]
    L.wf_psi_vjp_workspace_bytes.restype = i64
    L.wf_psi_vjp_workspace_bytes.argtypes = [vp, i64]
    L.wf_psi_vjp.restype = i32
    L.wf_psi_vjp.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.wf_logpdf_vjp_workspace_bytes.restype = i64
    L.wf_logpdf_vjp_workspace_bytes.argtypes = [vp, i64]
    L.wf_logpdf_vjp.restype = i32
    L.wf_logpdf_vjp.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp]
    L.wf_model_set_params_device.restype = i32
    L.wf_model_set_params_device.argtypes = [vp, vp, i64, vp]
    L.wf_adam_step.restype = i32
    L.wf_adam_step.argtypes = [vp, vp, vp, vp, i64, i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]
    L.wf_vqmc_train_step_workspace_bytes.restype = i64
    L.wf_vqmc_train_step_workspace_bytes.argtypes = [vp, i64]
    L.wf_vqmc_train_step.restype = i32
    L.wf_vqmc_train_step.argtypes = [vp, ctypes.POINTER(TrainState), ctypes.c_uint64, i64, vp, i32, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_float, ctypes.c_float, i32, vp, i64, vp]
    L.wf_nsc_workspace_bytes.restype = i64
    L.wf_nsc_workspace_bytes.argtypes = [i64, i32, i32]
    L.wf_nsc_fwd.restype = i32
    L.wf_nsc_fwd.argtypes = [vp, i64, i32, i32, ctypes.c_float, i32, vp, i32, vp, vp, vp, i64, vp]
    L.wf_vqmc_train_step_local.restype = i32
    L.wf_vqmc_train_step_local.argtypes = [vp, ctypes.POINTER(TrainState), ctypes.c_uint64, i64, vp, i32, ctypes.c_float, i32, vp, vp, i64, vp]
    L.wf_vqmc_train_step_apply.restype = i32
    L.wf_vqmc_train_step_apply.argtypes = [vp, ctypes.POINTER(TrainState), vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]
    L.wf_mle_train_step_workspace_bytes.restype = i64
    L.wf_mle_train_step_workspace_bytes.argtypes = [vp, i64]
    L.wf_mle_train_step.restype = i32
    L.wf_mle_train_step.argtypes = [vp, ctypes.POINTER(TrainState), vp, i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                    vp, i64, vp]
    L.wf_logpdf_loss_grad.restype = i32
    L.wf_logpdf_loss_grad.argtypes = [vp, vp, i64, ctypes.c_float, vp, vp, vp, i64, vp]
    L.wf_vqmc_loss_grad.restype = i32
    L.wf_vqmc_loss_grad.argtypes = [vp, vp, i64, vp, i32, ctypes.c_float, ctypes.c_float, vp, vp, vp, i64, vp]
    L.wf_vqmc_seeds.restype = i32
    L.wf_vqmc_seeds.argtypes = [vp, i64, i32, vp, i32, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
    L.wf_rqs_fwd.restype = i32
    L.wf_rqs_fwd.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                             vp, vp, vp, vp]
    L.wf_block_sums.restype = i32
    L.wf_block_sums.argtypes = [vp, i64, vp, vp, i64, vp]
    L.wf_block_sums_workspace_bytes.restype = i64
    L.wf_block_sums_workspace_bytes.argtypes = [i64]
    if L.wf_abi_version() != 2:
        raise ImportError("libwaveflow_hip ABI version mismatch")
    _lib = L
    return L


def check(status, what):
    if status < 0:
        raise WfError(status, what)
    return status
