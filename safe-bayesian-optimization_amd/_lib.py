"""ctypes binding of libsafebo.so (include/safebo.h).  No torch, no numpy C-API: plain pointers and sizes.

The library is the product's only compute path: if it is missing, or if no HIP device is present,
the calls raise -- there is no CPU fallback (the NumPy oracle under ``oracle/`` is test infrastructure
and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os

SBO_MAX_D = 8
SBO_MAX_Q = 8
SBO_MAX_N = 2048
SBO_COMM_ID_BYTES = 128

SBO_F64, SBO_F32 = 0, 1
SBO_MEAN, SBO_UCB, SBO_LCB, SBO_VAR = 0, 1, 2, 3
SBO_MASK_S, SBO_MASK_U, SBO_MASK_M, SBO_MASK_G, SBO_MASK_O = 0, 1, 2, 3, 4

SBO_OK = 0
SBO_E_INVALID = -1
SBO_E_NOMEM = -2
SBO_E_HIP = -3
SBO_E_NO_MODEL = -4
SBO_E_NO_CANDIDATES = -5
SBO_E_EMPTY_SAFE_SET = -6
SBO_E_COMM = -7
SBO_E_UNSUPPORTED = -8


class SweepOpts(C.Structure):
    _fields_ = [("b", C.c_double), ("reference_quirk_L_index", C.c_int32), ("want_masks", C.c_int32),
                ("posterior_ready", C.c_int32), ("lean", C.c_int32)]


class SafeOptResult(C.Structure):
    _fields_ = [
        ("minimizer_index", C.c_int64), ("minimizer_x", C.c_double * SBO_MAX_D), ("minimizer_std", C.c_double),
        ("expander_index_c", C.c_int64 * SBO_MAX_Q), ("expander_std_c", C.c_double * SBO_MAX_Q),
        ("expander_best_c", C.c_int32), ("expander_index", C.c_int64), ("expander_x", C.c_double * SBO_MAX_D),
        ("expander_std", C.c_double), ("choose_minimizer", C.c_int32), ("u_star", C.c_double),
        ("L", C.c_double * SBO_MAX_Q), ("count_S", C.c_int64), ("count_U", C.c_int64), ("count_M", C.c_int64),
        ("count_G", C.c_int64 * SBO_MAX_Q), ("n_exact_rechecks", C.c_int64),
        ("guard_band", C.c_int64), ("guard_rechecks", C.c_int64), ("guard_passes", C.c_int32), ("reserved_g", C.c_int32),
    ]


class GooseResult(C.Structure):
    _fields_ = [
        ("safe_min_index", C.c_int64), ("safe_min_x", C.c_double * SBO_MAX_D), ("safe_min_lcb", C.c_double),
        ("target_index_c", C.c_int64 * SBO_MAX_Q), ("target_lcb_c", C.c_double * SBO_MAX_Q),
        ("target_best_c", C.c_int32), ("target_index", C.c_int64), ("target_x", C.c_double * SBO_MAX_D),
        ("target_lcb", C.c_double), ("explore_index", C.c_int64), ("explore_x", C.c_double * SBO_MAX_D),
        ("choose_safe_min", C.c_int32), ("L", C.c_double * SBO_MAX_Q), ("count_S", C.c_int64),
        ("count_U", C.c_int64), ("count_O", C.c_int64 * SBO_MAX_Q), ("n_exact_rechecks", C.c_int64),
        ("guard_band", C.c_int64), ("guard_rechecks", C.c_int64), ("guard_passes", C.c_int32), ("reserved_g", C.c_int32),
    ]


class TRResult(C.Structure):
    _fields_ = [("index", C.c_int64), ("x", C.c_double * SBO_MAX_D), ("lcb", C.c_double), ("count_S", C.c_int64),
                ("count_T", C.c_int64), ("guard_band", C.c_int64), ("guard_rechecks", C.c_int64), ("guard_passes", C.c_int32),
                ("reserved_g", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [
        ("posterior_ms", C.c_double), ("classify_ms", C.c_double), ("expander_ms", C.c_double),
        ("argreduce_ms", C.c_double), ("comm_ms", C.c_double), ("total_ms", C.c_double),
        ("posterior_flops", C.c_double), ("candidates", C.c_int64), ("posterior_launches", C.c_int32),
        ("posterior_kernel", C.c_int32), ("posterior_executed_flops", C.c_double), ("posterior_setup_ms", C.c_double),
        ("fp64_rechecks", C.c_int64), ("recheck_ms", C.c_double),
        ("set_phase_ms", C.c_double), ("host_syncs", C.c_int32), ("comm_calls", C.c_int32), ("comm_bytes", C.c_int64),
        ("guard_dm", C.c_double * SBO_MAX_Q), ("guard_dv", C.c_double * SBO_MAX_Q), ("guard_rl", C.c_double * SBO_MAX_Q),
        ("guard_ms", C.c_double), ("halo_reruns", C.c_int32), ("set_path", C.c_int32),
        ("guard_audit_samples", C.c_int64), ("guard_audit_violations", C.c_int64), ("guard_audit_worst", C.c_double),
        ("guard_analytic_dm", C.c_double * SBO_MAX_Q), ("guard_analytic_dv", C.c_double * SBO_MAX_Q),
        ("guard_probe_dm", C.c_double * SBO_MAX_Q), ("guard_probe_dv", C.c_double * SBO_MAX_Q),
    ]


RELAY_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int)
RELAY_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

# every symbol include/safebo.h declares: (name, restype, argtypes)
_P = C.c_void_p
_DP = C.POINTER(C.c_double)
SYMBOLS = [
    ("sbo_version", C.c_int, []),
    ("sbo_last_error", C.c_char_p, []),
    ("sbo_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("sbo_init", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("sbo_shutdown", C.c_int, [_P]),
    ("sbo_synchronize", C.c_int, [_P]),
    ("sbo_comm_unique_id", C.c_int, [_P]),
    ("sbo_comm_init", C.c_int, [_P, C.c_int, C.c_int, _P]),
    ("sbo_comm_barrier", C.c_int, [_P]),
    ("sbo_comm_init_relay", C.c_int, [_P, C.c_int, C.c_int, RELAY_ALLREDUCE, RELAY_ALLGATHER, _P]),
    ("sbo_model_set", C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("sbo_model_set_list", C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("sbo_model_append", C.c_int, [_P, _P, _P]),
    ("sbo_candidates_points", C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, C.c_int64]),
    ("sbo_candidates_grid", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int64, C.c_int64]),
    ("sbo_candidates_grid_sharded", C.c_int, [_P, C.c_int, _P, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("sbo_posterior_run", C.c_int, [_P]),
    ("sbo_posterior_get", C.c_int, [_P, _P, _P]),
    ("sbo_bounds", C.c_int, [_P, C.c_double, C.c_int, C.c_int, _P]),
    ("sbo_sweep_safeopt", C.c_int, [_P, C.POINTER(SweepOpts), C.POINTER(SafeOptResult)]),
    ("sbo_sweep_goose", C.c_int, [_P, C.POINTER(SweepOpts), C.POINTER(GooseResult)]),
    ("sbo_sweep_tr", C.c_int, [_P, C.POINTER(SweepOpts), _P, C.c_double, C.POINTER(TRResult)]),
    ("sbo_explore_safeset", C.c_int, [_P, _P, C.POINTER(C.c_int64), _P]),
    ("sbo_masks_get", C.c_int, [_P, C.c_int, C.c_int, _P]),
    ("sbo_nll_batch", C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]),
    ("sbo_fit_de", C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, C.c_uint64, C.c_int, C.c_double, C.c_double, _P, _P,
                             C.POINTER(C.c_int)]),
    ("sbo_plant_wo", C.c_int, [_P, C.c_int64, _P, _P]),
    ("sbo_profile_get", C.c_int, [_P, C.POINTER(Profile)]),
    ("sbo_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
]


class SafeBOError(RuntimeError):
    """Library failure other than a bad argument (HIP / RCCL / memory)."""

    def __init__(self, code, msg):
        super().__init__(f"libsafebo error {code}: {msg}")
        self.code = code


class EmptySafeSetError(SafeBOError):
    """S_t is empty on the candidate set (the reference's DE would return an infeasible point)."""


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsafebo.so")


_lib = None


def load():
    """Load libsafebo.so (built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sbo_version() != 3:
        raise ImportError(f"libsafebo ABI version {lib.sbo_version()} != 3")
    _lib = lib
    return lib


def check(rc: int):
    """Status -> exception, keeping the reference's style: bad arguments are ValueError
    (models/GP_Safe.py:134-137, 159-162), everything else RuntimeError."""
    if rc == SBO_OK:
        return
    msg = load().sbo_last_error().decode("utf-8", "replace")
    if rc == SBO_E_INVALID:
        raise ValueError(msg)
    if rc == SBO_E_EMPTY_SAFE_SET:
        raise EmptySafeSetError(rc, msg)
    raise SafeBOError(rc, msg)
