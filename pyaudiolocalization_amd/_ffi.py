"""ctypes binding of libpal_hip.so (include/pal_hip.h).  No PyTorch, no CPU fallback: if the
library or a HIP device is missing every entry point raises - the product path never computes on
the host."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAL_LIB_PATH") or os.path.join(_HERE, "libpal_hip.so")   # (PAL_LIB_PATH: A/B builds, tools/build_variant.sh)

PAL_MAX_PEAKS = 256
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED, ERR_INTERNAL, ERR_COMM, ERR_MATERIAL = -1, -2, -3, -4, -5, -6, -7
BR_ALT_THRESHOLD, BR_ARGMAX_NO_PEAKS, BR_WINDOW_RETRY, BR_ARGMAX_WINDOW = 1, 2, 4, 8


class PhatParams(C.Structure):
    _fields_ = [("fs", C.c_double), ("threshold_multiplier", C.c_double), ("max_expected_delay", C.c_double),
                ("threshold_method", C.c_int32), ("peak_distance", C.c_int32), ("num_peaks", C.c_int32),
                ("reserved", C.c_int32)]


# one row of the TDOA table (pal_pair_record, 48 bytes)
RECORD = np.dtype([("k_sel", "<i4"), ("branch", "<i4"), ("k_argmax", "<i4"), ("n_sel", "<i4"),
                   ("cmax", "<f8"), ("cmin", "<f8"), ("snr", "<f8"), ("sel_height", "<f8")])
assert RECORD.itemsize == 48


class PalError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"libpal_hip error {code}: {text}")
        self.code = code
        self.text = text


_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol of include/pal_hip.h
SIGNATURES = {
    "pal_abi_version": (C.c_int, []),
    "pal_create": (C.c_int, [C.c_int, C.POINTER(_H)]),
    "pal_destroy": (None, [_H]),
    "pal_last_error": (C.c_char_p, [_H]),
    "pal_synchronize": (C.c_int, [_H]),
    "pal_clear_plans": (C.c_int, [_H]),
    "pal_set_max_plans": (C.c_int, [_H, C.c_int]),
    "pal_plan_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pal_set_chunk": (C.c_int, [_H, C.c_int]),
    "pal_pair_group_size": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int32)]),
    "pal_device_alloc": (C.c_int, [_H, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pal_device_free": (C.c_int, [_H, C.c_void_p]),
    "pal_upload": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pal_download": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pal_gcc_phat_all_pairs": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(PhatParams), C.c_void_p,
                                         C.c_void_p]),
    "pal_gcc_phat_all_pairs_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(PhatParams), C.c_void_p]),
    "pal_gcc_phat_pairs": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(PhatParams), C.c_void_p]),
    "pal_gcc_phat_pairs_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(PhatParams), C.c_void_p]),
    "pal_phat_correlation": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pal_get_time_delays_phat": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(PhatParams), C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "pal_corr_metrics": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p]),
    "pal_image_sources": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_double, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_int)]),
    "pal_simulate_multipath": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pal_fractional_delay": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p]),
    "pal_normalize_compress": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "pal_filtfilt": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_void_p]),
    "pal_wiener3": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pal_xcorr_vs_ref": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_double)]),
    "pal_simulate_multipath_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pal_row_energies_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pal_sync_measure_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "pal_align_rows_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pal_filtfilt_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p]),
    "pal_wiener3_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pal_filtfilt_ragged_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "pal_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pal_comm_init": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "pal_comm_all_gather": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pal_comm_destroy": (C.c_int, [_H]),
    "pal_profile_begin": (C.c_int, [_H]),
    "pal_profile_end": (C.c_int, [_H]),
    "pal_profile_sampling": (C.c_int, [_H, C.c_int]),
    "pal_profile_get": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pal_profile_entry": (C.c_int, [_H, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pal_plan_info": (C.c_int, [_H, C.c_int, _PI, _PI, _PI, _PI]),
    "pal_plan_factors": (C.c_int, [_H, C.c_int, _PI, _PI, _PI]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libpal_hip.so and bind every declared symbol; raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C pyaudiolocalization_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the build lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None else a.ctypes.data


def f64(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and out.shape != tuple(shape):
        raise ValueError(f"expected array of shape {tuple(shape)}, got {out.shape}")
    return out
