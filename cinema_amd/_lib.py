"""ctypes binding of libcinema_hip.so (the C ABI declared in include/cinema_hip.h).

The library is built in-tree by `cinema_amd.build.build()` (hipcc, gfx950).  Loading fails
loudly when the shared object is missing: there is no CPU or eager-PyTorch fallback for the
hot path.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_LAYERS = 16
PREC_F32, PREC_BF16 = 0, 1
BWD_SKIP_WEIGHT_GRADS = 1
BWD_GRADS_CLEARED, BWD_DCOORDS_CLEARED = 2, 64
BWD_WEIGHTS_FIRST, BWD_DLAT_ONLY = 256, 512
NODES_ROUND_S = 128
NODE_SLOTS = 32
PREPARE_SCRATCH = 260
PENDING_BYTES = 4096
NOALLOC_MAX_N, NOALLOC_MAX_S = 65536, 4096
BWD_PHASE_CHAIN, BWD_PHASE_DLAT, BWD_PHASE_DW, BWD_PHASE_FINAL_DW = 4, 8, 16, 32

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CINEMA_HIP_LIB") or os.path.join(_HERE, "libcinema_hip.so")   # override: developer builds


class SineLayerDesc(C.Structure):
    """Mirror of `cinema_sine_layer_desc` (include/cinema_hip.h)."""
    _fields_ = [("in_feat", C.c_int32), ("lat_feat", C.c_int32), ("out_feat", C.c_int32), ("precision", C.c_int32),
                ("omega", C.c_float)]


class SirenDesc(C.Structure):
    """Mirror of `cinema_siren_desc` (include/cinema_hip.h)."""
    _i64L = C.c_int64 * MAX_LAYERS
    _i32L = C.c_int32 * MAX_LAYERS
    _fields_ = [
        ("n_layers", C.c_int32), ("in_dim", C.c_int32), ("hidden", C.c_int32), ("lat", C.c_int32),
        ("out_dim", C.c_int32), ("precision", C.c_int32), ("modulated_mask", C.c_uint32),
        ("omega_first", C.c_float), ("omega_hidden", C.c_float),
        ("HP", C.c_int32), ("LP", C.c_int32), ("n_saved", C.c_int32), ("n_gbuf", C.c_int32),
        ("off_w", _i64L), ("off_b", _i64L), ("off_wl", _i64L), ("off_bl", _i64L),
        ("off_wf", C.c_int64), ("off_bf", C.c_int64), ("n_params", C.c_int64),
        ("pk_w", _i64L), ("pk_wt", _i64L), ("pk_wls", _i64L), ("pk_wlt", _i64L), ("pk_wlsT", _i64L), ("pk_wltT", _i64L),
        ("pk_wf", C.c_int64), ("pk_total", C.c_int64),
        ("sv_y", _i32L), ("sv_c", _i32L), ("sv_z", _i32L), ("sv_s", _i32L),
        ("gb_z", _i32L), ("gb_s", _i32L), ("gb_t", _i32L),
        ("pk_fs", C.c_int64), ("pk_bs", C.c_int64), ("pk_side", C.c_int64), ("fs_steps", C.c_int32), ("bs_steps", C.c_int32), ("gb_extra_bytes", C.c_int64),
        ("pk_ns", C.c_int64), ("pk_nb", C.c_int64), ("ns_steps", C.c_int32), ("nb_steps", C.c_int32),
    ]


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_F = C.c_float

# name -> argtypes; every function returns int (0 = ok) unless listed in _RESTYPES
SIGNATURES = {
    "cinema_version": [],
    "cinema_last_error": [],
    "cinema_siren_plan": [C.POINTER(SirenDesc)],
    "cinema_siren_pack": [C.POINTER(SirenDesc), _P, _P, _P],
    "cinema_siren_pack_part": [C.POINTER(SirenDesc), _P, _P, _I32, _P],
    "cinema_siren_forward": [C.POINTER(SirenDesc), _P, _P, _P, _P, _I64, _I64, _P, _P, _P],
    "cinema_siren_backward": [C.POINTER(SirenDesc), _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _I32, _P],
    "cinema_siren_backward_deferred": [C.POINTER(SirenDesc), _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _I32, _P, _P],
    "cinema_affine_from_embedding": [_P, _I64, _I32, _P, _P],
    "cinema_latent_to_node_major": [_P, _I64, _I32, _I32, _P, _P],
    "cinema_affine_trilerp_fwd": [_P, _P, _P, _I32, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P,
                                  _I32, _P],
    "cinema_affine_trilerp_fwd_sorted": [_P, _P, _P, _I32, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P,
                                         _P, _I32, _P],
    "cinema_cond_table_bwd": [_P, _P, _I64, _I64, _I32, _P, _P],
    "cinema_step_head": [C.POINTER(SirenDesc), _P, _P, _P, _P, _P, _I32, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32,
                         _I32, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I32, _P, _P, _P],
    "cinema_affine_trilerp_bwd": [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32,
                                  _I32, _I32, _P, _P, _P, _P, _P],
    "cinema_affine_trilerp_bwd_noalloc": [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32,
                                          _I32, _I32, _P, _P, _P, _P],
    "cinema_affine_trilerp_bwd_sorted": [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32,
                                         _I32, _I32, _P, _P, _P, _P, _P],
    "cinema_affine_trilerp_bwd_sorted_pending": [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32,
                                                 _I32, _I32, _I32, _P, _P, _P, _P, _P, _P],
    "cinema_nodes_sizes": [C.POINTER(SirenDesc), _I64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "cinema_node_weights": [_P, _P, _P, _I32, _I64, _I64, _I64, _I32, _I32, _I32, _P, _P, _P],
    "cinema_node_tiles": [_P, _I64, _I64, _I64, _P, _P],
    "cinema_node_tables": [C.POINTER(SirenDesc), _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P],
    "cinema_siren_forward_nodes": [C.POINTER(SirenDesc), _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _I32, _P],
    "cinema_siren_backward_nodes": [C.POINTER(SirenDesc), _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _P, _P, _P, _P,
                                    _I32, _P, _P],
    "cinema_nodegrad_eligible": [C.POINTER(SirenDesc), _I32, _I32, _I32],
    "cinema_nodegrad_work_bytes": [C.POINTER(SirenDesc), _I64, _I64],
    "cinema_siren_forward_node_prep": [C.POINTER(SirenDesc), _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "cinema_nodegrad_prepare": [C.POINTER(SirenDesc), _P, _P, _P, _P, _I32, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32,
                                _P, _I32, _P],
    "cinema_siren_backward_nodegrad": [C.POINTER(SirenDesc), _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _I32, _P, _P, _P, _I32,
                                       _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "cinema_sample_loss_fwd_bwd": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _P, _F, _F, _P, _P, _P, _P],
    "cinema_sample_loss_fwd_bwd_prepared": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _P, _F, _F, _P, _P, _P, _P],
    "cinema_sample_loss_fwd_bwd_normed": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _P, _F, _F, _P, _P, _P, _P, _P],
    "cinema_step_prepare": [_P, _I64, _P, _P, _P, _P, _I64, _I64, _I32, _P, _P, _P],
    "cinema_conv_mod_work_bytes": [_I64, _I32, _I32, _I32, _I32, _I32, _I32],
    "cinema_conv_mod_forward": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P],
    "cinema_conv_mod_backward": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "cinema_sine_layer_work_bytes": [_P, _I64],
    "cinema_sine_layer_forward": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P],
    "cinema_sine_layer_backward": [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "cinema_infer_minmax": [_P, _I64, _I32, _I32, _P, _P],
    "cinema_infer_finish": [_P, _I64, _I32, _I32, _P, _P, _P],
    "cinema_infer_chunk": [_P, _I64, _I32, _I32, _P, _P, _P],
    "cinema_cc_label": [_P, _I32, _I32, _I32, _P, _P, _I32, _I32, _P],
    "cinema_infer_rescale": [_P, _I64, _I32, _I32, _P, _P],
    "cinema_grid_coords": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P],
}
_RESTYPES = {"cinema_last_error": C.c_char_p, "cinema_sine_layer_work_bytes": C.c_int64, "cinema_conv_mod_work_bytes": C.c_int64,
             "cinema_nodegrad_work_bytes": C.c_int64}

_lib = None


class CinemaHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CinemaHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The hot path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().cinema_last_error()
        raise CinemaHipError(f"{what or 'libcinema_hip'} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


ADAMW_MAX_GROUPS = 8


class AdamWGroup(C.Structure):
    """Mirror of `cinema_adamw_group` (include/cinema_hip.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step", C.c_int32)]


SIGNATURES["cinema_adamw_step"] = [C.POINTER(AdamWGroup), _I32, _P]
SIGNATURES["cinema_adamw_step_clear"] = [C.POINTER(AdamWGroup), _I32, _P]
