"""ctypes binding of libsvgpfa_hip.so (the C ABI declared in include/svgpfa.h).

There is NO fallback: if the HIP library is missing or a symbol is absent, importing the
engine fails loudly.  Build it with `python __graft_entry__.py` (or `make -C svgpfa_amd/csrc`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVGPFA_LIB: diagnosis builds of the same library (tests/tools), never a different implementation
LIB_PATH = os.environ.get("SVGPFA_LIB") or os.path.join(_HERE, "csrc", "libsvgpfa_hip.so")

OK, ERR_INVALID, ERR_NOT_PD, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = range(6)
KERNEL_RBF, KERNEL_PERIODIC = 0, 1
GRAD_VARIATIONAL, GRAD_EMBEDDING, GRAD_KERNELS, GRAD_INDPOINTS = 1, 2, 4, 8
GRAD_ALL = 15
PRECISION_FP64, PRECISION_MIXED = 0, 1


class Dims(ctypes.Structure):
    _fields_ = [("n_trials", ctypes.c_int32), ("n_neurons", ctypes.c_int32),
                ("n_latents", ctypes.c_int32), ("n_quad", ctypes.c_int32),
                ("n_ind", ctypes.POINTER(ctypes.c_int32)),
                ("kernel_type", ctypes.POINTER(ctypes.c_int32)),
                ("scale", ctypes.POINTER(ctypes.c_double)),
                ("lengthscale_scale", ctypes.POINTER(ctypes.c_double)),
                ("period_scale", ctypes.POINTER(ctypes.c_double)),
                ("precision", ctypes.c_int32)]


_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)
_i32, _i64, _u32, _f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint, ctypes.c_double

# name -> (restype, argtypes): every symbol include/svgpfa.h declares
SIGNATURES = {
    "svgpfa_create": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.c_int, ctypes.POINTER(_vp)]),
    "svgpfa_destroy": (None, [_vp]),
    "svgpfa_last_error": (ctypes.c_char_p, [_vp]),
    "svgpfa_set_stream": (ctypes.c_int, [_vp, _vp]),
    "svgpfa_sync": (ctypes.c_int, [_vp]),
    "svgpfa_set_reg": (ctypes.c_int, [_vp, _f64]),
    "svgpfa_set_quad": (ctypes.c_int, [_vp, _vp, _vp]),
    "svgpfa_set_spikes": (ctypes.c_int, [_vp, ctypes.POINTER(_i64), _vp, _vp]),
    "svgpfa_build_kernels": (ctypes.c_int, [_vp, _vpp, _vpp]),
    "svgpfa_shared_len": (_i64, [_vp]),
    "svgpfa_elbo": (ctypes.c_int, [_vp, _vpp, _vpp, _vp, _vp, _u32, _vp, _vpp, _vpp, _vpp]),
    "svgpfa_latent_stats": (ctypes.c_int, [_vp, _vpp, _vpp, _vp, _vp, _vp, _vp]),
    "svgpfa_latent_stats_backward": (ctypes.c_int, [_vp, _vpp, _vpp, _vp, _vp, _vp, _vp, _u32, _vp, _vpp, _vpp, _vpp]),
    "svgpfa_freeze_stats": (ctypes.c_int, [_vp, _vpp, _vpp, _vp]),
    "svgpfa_kl": (ctypes.c_int, [_vp, _vp]),
    "svgpfa_get_series_harmonics": (ctypes.c_int, [_vp, ctypes.POINTER(_i32)]),
    "svgpfa_ell_given_stats": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp]),
    "svgpfa_predict": (ctypes.c_int, [_vp, _vpp, _vpp, _vp, _i32, _vp, _vp]),
    "svgpfa_get_kzz": (ctypes.c_int, [_vp, _i32, _vp]),
    "svgpfa_get_kzz_chol": (ctypes.c_int, [_vp, _i32, _vp]),
    "svgpfa_get_ktz_quad": (ctypes.c_int, [_vp, _i32, _vp]),
    "svgpfa_get_ktz_spikes": (ctypes.c_int, [_vp, _i32, _vp]),
    "svgpfa_set_embedding_token": (ctypes.c_int, [_vp, ctypes.c_uint64]),
    "svgpfa_enable_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "svgpfa_kernel_ms": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_f64), ctypes.POINTER(_i32)]),
    "svgpfa_spike_kernel_ms": (ctypes.c_int, [_vp, ctypes.POINTER(_f64), ctypes.POINTER(_i32)]),
    "svgpfa_lbfgs_workspace_doubles": (_i64, [_i64, _i32]),
    "svgpfa_lbfgs_coef": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "svgpfa_lbfgs_grad_stats": (ctypes.c_int, [_vp, _i64, _vp, _f64, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp]),
    "svgpfa_lbfgs_pair_dots": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _f64, _vp, _vp, _i64, _i32, _i64, _f64, _vp, _vp]),
    "svgpfa_lbfgs_direction": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load the in-tree HIP library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the svGPFA HIP engine is not built. "
            "Run `python __graft_entry__.py` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # The engine works on torch's device memory and streams, so it must share torch's HIP runtime: import torch FIRST.
    # Loaded the other way round, this library binds the system libamdhip64 and torch then brings its own copy -- two
    # runtimes in one process, and the first one reports "no ROCm-capable device".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
