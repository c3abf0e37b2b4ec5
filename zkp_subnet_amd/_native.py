"""ctypes binding of libkzg_mi355x.so (include/kzg_mi355x.h).  The HIP library IS the product: if it is
missing or no gfx950 device works, everything here raises -- there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# Eight hardware queues instead of the HIP runtime's four, unless the user chose: the four lanes of a context then run
# concurrently whatever the creation order of the process's streams (measured: profiles/r05_ab_hw_queues.log).  The runtime
# reads the variable once, at its first call -- so it is set HERE, when this module is imported: before the package starts
# a thread or touches HIP.  The library itself never writes the environment; it measures what it got (kzg_runtime_info) and
# `Client.start` warns when the lanes do not overlap (HIP was initialised earlier by somebody else, or the user chose fewer).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# KZG_MI355X_LIB: another BUILD of the same library (A/B candidates under zkp_subnet_amd/ab/, the prototype build): dev
# scripts point at it instead of overwriting the shipped file.  Never a different implementation, never a fallback.
LIB_PATH = os.environ.get("KZG_MI355X_LIB") or os.path.join(HERE, "libkzg_mi355x.so")

KZG_OK, KZG_E_ARG, KZG_E_SCALAR, KZG_E_POINT, KZG_E_HIP, KZG_E_NOMEM, KZG_E_BUSY, KZG_E_COMM = 0, -1, -2, -3, -4, -5, -6, -7
STATUS_NAMES = {0: "OK", -1: "E_ARG", -2: "E_SCALAR", -3: "E_POINT", -4: "E_HIP", -5: "E_NOMEM", -6: "E_BUSY", -7: "E_COMM"}
TIMING_NAMES = ["decode", "ntt", "digits", "scan", "scatter", "accumulate", "fixup", "tree", "final", "poly", "total", "collective"]

# every symbol include/kzg_mi355x.h (serving surface) and include/kzg_mi355x_test.h (test hooks) declare:
# name -> (restype, argtypes)
_P = ctypes.c_void_p
_B = ctypes.c_char_p
_U64, _U32, _I = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
SYMBOLS = {
    "kzg_create": (_I, [_I, ctypes.POINTER(_P)]),
    "kzg_destroy": (None, [_P]),
    "kzg_last_error": (ctypes.c_char_p, [_P]),
    "kzg_version": (ctypes.c_char_p, []),
    "kzg_runtime_info": (_I, [_P, ctypes.POINTER(ctypes.c_int32)]),
    "kzg_set_window": (_I, [_P, _I]),
    "kzg_get_window": (_I, [_P]),
    "kzg_get_window_layout": (_I, [_P, ctypes.POINTER(ctypes.c_int32), _I]),
    "kzg_load_srs": (_I, [_P, _B, _U64, _I, _I]),
    "kzg_load_srs_compressed": (_I, [_P, _B, _U64, _I, _I]),
    "kzg_load_srs_file": (_I, [_P, _B, _I, _I, _I]),
    "kzg_load_srs_file_slices": (_I, [_P, _B, _I, _I, _I, _U32, _U32]),
    "kzg_load_srs_file_range": (_I, [_P, _B, _I, _U64, _U64, _I]),
    "kzg_get_load_stats": (_I, [_P, ctypes.POINTER(ctypes.c_double)]),
    "kzg_set_srs_subgroup_check": (_I, [_P, _I]),
    "kzg_gen_srs": (_I, [_P, _B, _B, _U32, _I, _I]),
    "kzg_srs_points": (_U64, [_P]),
    "kzg_srs_read": (_I, [_P, _I, _U64, _U64, _B]),
    "kzg_srs_read_compressed": (_I, [_P, _I, _U64, _U64, _B]),
    "kzg_commit": (_I, [_P, _U32, _B, _U64, _I, _B]),
    "kzg_open": (_I, [_P, _U32, _B, _U64, _I, _B, _B, _B]),
    "kzg_commit_open": (_I, [_P, _U32, _B, _U64, _I, _B, _B, _B, _B]),
    "kzg_commit_cached": (_I, [_P, _U32, _B, _U64, _I, _B, _B]),
    "kzg_open_cached": (_I, [_P, _U32, _B, _U64, _I, _B, _B, _B, _B]),
    "kzg_row_cache_stats": (_I, [_P, ctypes.POINTER(_U64)]),
    "kzg_msm": (_I, [_P, _B, _U64, _U64, _B]),
    "kzg_ntt": (_I, [_P, _B, _U64, _I]),
    "kzg_eval": (_I, [_P, _B, _U64, _B, _B]),
    "kzg_ntt_eval": (_I, [_P, _B, _U64, _I, _B, _B]),
    "kzg_vk_create": (_I, [_B, _B, _U32, ctypes.POINTER(_P)]),
    "kzg_vk_create_synthetic": (_I, [_B, _B, _U32, ctypes.POINTER(_P)]),
    "kzg_vk_destroy": (None, [_P]),
    "kzg_vk_export": (_I, [_P, _B, _U64]),
    "kzg_vk_verify": (_I, [_P, _U32, _B, _B, _B, _B, ctypes.POINTER(_I)]),
    "kzg_vk_verify_batch": (_I, [_P, _U32, ctypes.POINTER(_U32), _B, _B, _B, _B, _I, ctypes.POINTER(_I)]),
    "kzg_vk_pairing": (_I, [_B, _B, _B]),
    "kzg_msm_partial": (_I, [_P, _B, _U64, _U64, _B]),
    "kzg_g1_sum": (_I, [_P, _B, _U32, _B]),
    "kzg_msm_partial_resident_dev": (_I, [_P, _I, _U64, _U64, _P]),
    "kzg_g1_sum_dev": (_I, [_P, _P, _U32, _B]),
    "kzg_msm_sharded_begin": (_I, [_P, _I, _U64, _U64, _P, _P, ctypes.POINTER(_I)]),
    "kzg_msm_sharded_finish": (_I, [_P, _I, _P, _U32, _P, _B]),
    "kzg_comm_unique_id": (_I, [_B]),
    "kzg_comm_init": (_I, [_P, _B, _I, _I]),
    "kzg_comm_init_bounded": (_I, [_P, _B, _I, _I, _I]),
    "kzg_comm_destroy": (_I, [_P]),
    "kzg_comm_set_timeout": (_I, [_P, _I]),
    "kzg_comm_info": (_I, [_P, ctypes.POINTER(ctypes.c_int32)]),
    "kzg_comm_selftest": (_I, [_P]),
    "kzg_msm_sharded": (_I, [_P, _I, _U64, _U64, _B]),
    "kzg_test_comm_stall": (_I, [_P, _I]),
    "kzg_test_comm_stall_n": (_I, [_P, _I, _I]),
    "kzg_multi_create": (_I, [_I, ctypes.POINTER(_I), ctypes.POINTER(_P)]),
    "kzg_multi_destroy": (None, [_P]),
    "kzg_multi_last_error": (ctypes.c_char_p, [_P]),
    "kzg_multi_count": (_I, [_P]),
    "kzg_multi_ctx": (_P, [_P, _I]),
    "kzg_multi_device_of": (_I, [_P, _U32]),
    "kzg_multi_load_srs_file": (_I, [_P, _B, _I, _I, _I]),
    "kzg_multi_gen_srs": (_I, [_P, _B, _B, _I, _I]),
    "kzg_multi_load_srs_file_segments": (_I, [_P, _B, _I, _U64]),
    "kzg_multi_gen_srs_segments": (_I, [_P, _B, _B, _U64]),
    "kzg_multi_segment": (_I, [_P, _I, ctypes.POINTER(_U64)]),
    "kzg_multi_msm": (_I, [_P, _B, _U64, _U64, _B]),
    "kzg_multi_upload_fr": (_I, [_P, _I, _B, _U64, _U64]),
    "kzg_multi_msm_resident": (_I, [_P, _I, _B]),
    "kzg_multi_commit": (_I, [_P, _U32, _B, _U64, _I, _B]),
    "kzg_multi_open": (_I, [_P, _U32, _B, _U64, _I, _B, _B, _B]),
    "kzg_multi_commit_open": (_I, [_P, _U32, _B, _U64, _I, _B, _B, _B, _B]),
    "kzg_multi_commit_open_rows": (_I, [_P, _U32, ctypes.POINTER(_U32), _B, _U64, _I, _B, _B, _B, _B, ctypes.POINTER(_I)]),
    "kzg_upload_fr": (_I, [_P, _I, _B, _U64, _I]),
    "kzg_msm_resident": (_I, [_P, _I, _U64, _U64, _B]),
    "kzg_msm_partial_resident": (_I, [_P, _I, _U64, _U64, _B]),
    "kzg_msm_submit": (_I, [_P, _I, _U64, _U64, _I, ctypes.POINTER(_I)]),
    "kzg_msm_wait": (_I, [_P, _I, _B]),
    "kzg_msm_cancel": (_I, [_P, _I]),
    "kzg_commit_open_resident": (_I, [_P, _U32, _I, _U64, _I, _B, _B, _B, _B]),
    "kzg_ntt_resident": (_I, [_P, _I, _U64, _I]),
    "kzg_staging_acquire": (_I, [_P, _U64, ctypes.POINTER(_P), ctypes.POINTER(_I)]),
    "kzg_staging_release": (_I, [_P, _I]),
    "kzg_staging_flush": (_I, [_P, _I, _U64, _U64]),
    "kzg_set_host_finish": (_I, [_P, _I]),
    "kzg_host_xyzz_to_c48": (_I, [_P, _B]),
    "kzg_host_xyzz_pair_to_c48": (_I, [_P, _P, _B, _B]),
    "kzg_host_xyzz_to_partial192": (_I, [_P, _B]),
    "kzg_g1_sum_compressed": (_I, [_P, _B, _U32, _B]),
    "kzg_set_profiling": (_I, [_P, _I]),
    "kzg_get_timings": (_I, [_P, ctypes.POINTER(ctypes.c_float), _I]),
    "kzg_msm_plan": (_I, [_P, _U64, ctypes.POINTER(ctypes.c_int32)]),
    "kzg_calibrate": (_I, [_P, _I, ctypes.POINTER(ctypes.c_double)]),
    "kzg_b64_decode_fr": (_I, [_B, _U64, _B]),
    "kzg_b64_encode_fr": (_I, [_B, _U64, _B]),
    "kzg_test_field": (_I, [_P, _I, _I, _B, _B, _B, _U64]),
    "kzg_test_g1": (_I, [_P, _I, _B, _B, _B, _U64]),
}

_lib = None


class KzgError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"kzg_mi355x {STATUS_NAMES.get(code, code)}: {message}")
        self.code = code


def lib_available() -> bool:
    return os.path.exists(LIB_PATH)


def load() -> ctypes.CDLL:
    """Loads the HIP library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m zkp_subnet_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
