"""ctypes binding of libneupan_amd.so (C ABI: include/neupan_amd.h).

The HIP library is the product path.  There is NO CPU fallback: if the shared library is
missing (or cannot be loaded) every entry point raises -- build it with
`python -m neupan_amd.build` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (NPA_LIB_PATH: a variant build of the same library -- the experiments build, a profiling build -- for tools and tests; there is
# no fallback of any kind: whichever path is named must exist and export every symbol of include/neupan_amd.h)
LIB_PATH = os.environ.get("NPA_LIB_PATH") or os.path.join(HERE, "libneupan_amd.so")

NPA_MAX_T, NPA_MAX_M, NPA_MAX_E = 21, 32, 8
KIN = {"diff": 0, "acker": 1, "omni": 2}


class NpaConfig(C.Structure):
    _fields_ = [
        ("receding", C.c_int32), ("iter_num", C.c_int32), ("dune_max_num", C.c_int32),
        ("nrmp_max_num", C.c_int32), ("edge_num", C.c_int32), ("kinematics", C.c_int32),
        ("iter_threshold", C.c_float),
        ("step_time", C.c_double), ("wheelbase", C.c_double),
        ("speed_bound", C.c_double * 2), ("acce_bound", C.c_double * 2),
        ("ro_obs", C.c_double), ("bk", C.c_double),
        ("q_s", C.c_float * 3), ("p_u", C.c_float), ("eta", C.c_float), ("d_max", C.c_float), ("d_min", C.c_float),
        ("G", (C.c_float * 2) * NPA_MAX_E), ("h", C.c_float * NPA_MAX_E),
    ]


class NpaForwardCall(C.Structure):
    """npa_forward_call (include/neupan_amd.h): one call of a breadth-first burst, npa_forward_batch_group."""
    _fields_ = [("h", C.c_void_p), ("batch", C.c_int32), ("n_stride", C.c_int32), ("iter_num", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points", "velocities", "n_points", "out_s", "out_u",
                                          "out_d", "out_min_distance", "out_iters", "out_nrmp_points")] + \
               [("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("state", C.c_void_p), ("state_bytes", C.c_size_t),
                ("stream", C.c_void_p)]


class NpaDuneWeights(C.Structure):
    _fields_ = [("lin_w", C.c_void_p * 6), ("lin_b", C.c_void_p * 6), ("ln_w", C.c_void_p * 3), ("ln_b", C.c_void_p * 3)]


# every symbol include/neupan_amd.h declares: (name, restype, argtypes)
_P, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "npa_create": (_I, [C.POINTER(NpaConfig), C.POINTER(NpaDuneWeights), C.POINTER(_P)]),
    "npa_destroy": (_I, [_P]),
    "npa_key_mode": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "npa_geo_report": (_I, [_P, C.POINTER(C.c_float), _I]),
    "npa_audit_read": (_I, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float), _I]),
    "npa_audit_peek": (_I, [_P, C.POINTER(C.c_uint64)]),
    "npa_use_network_keys": (_I, [_P]),
    "npa_selftest_flags": (_I, [_P, C.POINTER(C.c_int)]),
    "npa_pack_cache_stats": (_I, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "npa_set_adjust": (_I, [_P, C.POINTER(C.c_float * 3), C.c_float, C.c_float, C.c_float, C.c_float]),
    "npa_workspace_bytes": (_SZ, [_P, _I]),
    "npa_state_bytes": (_SZ, [_P, _I]),
    "npa_workspace_qp_info_offset": (_SZ, [_P, _I]),
    "npa_workspace_layout": (_I, [_P, _I, C.POINTER(C.c_size_t), _I]),
    "npa_forward_batch": (_I, [_P, _I, _I] + [_P] * 13 + [_P, _SZ, _P, _SZ, _P]),
    "npa_forward_begin": (_I, [_P, _I, _I] + [_P] * 13 + [_P, _SZ, _P, _SZ, _P, _I]),
    "npa_forward_batch_flags": (_I, [_P, _I, _I] + [_P] * 13 + [_P, _SZ, _P, _SZ, _P, _I]),
    "npa_forward_iter": (_I, [_P, _I]),
    "npa_forward_end": (_I, [_P]),
    "npa_forward_batch_group": (_I, [_I, C.POINTER(NpaForwardCall), _I]),
    "npa_forward_group_merged": (_I, [_I, C.POINTER(NpaForwardCall)]),
    "npa_dune_stage": (_I, [_P, _I, _I] + [_P] * 9 + [_P]),
    "npa_nrmp_stage": (_I, [_P, _I] + [_P] * 13 + [_P]),
    "npa_nrmp_params": (_I, [_P, _I] + [_P] * 8 + [_P]),
    "npa_nrmp_backward": (_I, [_P, _I] + [_P] * 17 + [_P]),
    "npa_nominal_ref_states": (_I, [_I, _I, _I, C.c_double, C.c_double] + [_P] * 12 + [_P]),
    "npa_path_progress": (_I, [_I, _P, _P, _P, _P, _P, C.c_double, _I, C.c_double, _I, _P, _P, _P]),
    "npa_scan_to_points": (_I, [_I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "npa_dune_labels": (_I, [_I, _P, _P, C.c_int64, _P, _P, _P, _P]),
    "npa_profile_enable": (_I, [_P, _I]),
    "npa_profile_read": (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "npa_profile_read_aset": (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "npa_last_error": (C.c_char_p, []),
    "npa_version": (C.c_char_p, []),
}

_lib = None


class NeupanAmdError(RuntimeError):
    pass


def load():
    """Load the HIP library; raise loudly if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: its wheel bundles the HIP runtime (libamdhip64) that owns
    # the device context and the streams we are handed; loading ours after it makes the
    # dynamic linker bind this library to that same runtime instead of a second copy.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NeupanAmdError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m neupan_amd.build` "
            "(needs hipcc; cross-compiles for gfx950). neupan_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().npa_last_error()
        raise NeupanAmdError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
