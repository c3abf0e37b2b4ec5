"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- ctypes binding of oracle/kzg_cpu.c.

The C restatement is the mid-size checker (sizes the pure-Python oracle cannot reach in
seconds) and the reported CPU baseline ("cpu_baseline.kind": "port").  PARITY UNPINNED
against the real `fourier` prover; pinned by the reference Fr KAT and by agreement with
oracle/bls12_381.py on tests/golden/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkzg_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "kzg_cpu.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libkzg_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        u8p, u64, i32 = ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int
        _lib.orc_g1_mul_gen.argtypes = [u8p, u8p]
        _lib.orc_g1_compress.argtypes = [u8p, u8p]
        _lib.orc_g1_sum.argtypes = [u8p, u64, u8p]
        _lib.orc_fr_ntt.argtypes = [u8p, u64, i32]
        _lib.orc_fr_eval.argtypes = [u8p, u64, u8p, u8p]
        _lib.orc_srs_gen.argtypes = [u8p, u8p, i32, i32, ctypes.c_uint32, u8p]
        _lib.orc_msm.argtypes = [u8p, u8p, u64, i32, u8p]
        _lib.orc_msm_prepare.argtypes = [u8p, u8p, u64]
        _lib.orc_msm_prepare.restype = ctypes.c_void_p
        _lib.orc_msm_run.argtypes = [ctypes.c_void_p, i32, u8p]
        _lib.orc_msm_free.argtypes = [ctypes.c_void_p]
        _lib.orc_commit.argtypes = [u8p, u8p, u64, i32, i32, u8p]
        _lib.orc_open.argtypes = [u8p, u8p, u64, i32, u8p, i32, u8p, u8p]
        _lib.orc_usable_cpus.restype = i32
    return _lib


def _chk(rc: int, what: str) -> None:
    if rc != 0:
        raise ValueError(f"oracle {what} failed rc={rc}")


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask cut down to the cgroup quota (thread counts above it are capped)."""
    return int(lib().orc_usable_cpus())


def g1_mul_gen(k_be32: bytes) -> bytes:
    out = ctypes.create_string_buffer(48)
    _chk(lib().orc_g1_mul_gen(k_be32, out), "g1_mul_gen")
    return out.raw


def g1_compress(pt_be96: bytes) -> bytes:
    out = ctypes.create_string_buffer(48)
    _chk(lib().orc_g1_compress(pt_be96, out), "g1_compress")
    return out.raw


def g1_sum(pts_be96: bytes) -> bytes:
    out = ctypes.create_string_buffer(48)
    _chk(lib().orc_g1_sum(pts_be96, len(pts_be96) // 96, out), "g1_sum")
    return out.raw


def fr_ntt(vals_be32: bytes, inverse: bool) -> bytes:
    buf = ctypes.create_string_buffer(vals_be32, len(vals_be32))
    _chk(lib().orc_fr_ntt(buf, len(vals_be32) // 32, int(inverse)), "fr_ntt")
    return buf.raw


def fr_eval(coeffs_be32: bytes, x_be32: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    _chk(lib().orc_fr_eval(coeffs_be32, len(coeffs_be32) // 32, x_be32, out), "fr_eval")
    return out.raw


def srs_gen(tau_x_be32: bytes, tau_y_be32: bytes, scale: int, machines_scale: int, i: int) -> bytes:
    T = 1 << (scale - machines_scale)
    out = ctypes.create_string_buffer(96 * T)
    _chk(lib().orc_srs_gen(tau_x_be32, tau_y_be32, scale, machines_scale, i, out), "srs_gen")
    return out.raw


def msm(points_be96: bytes, scalars_be32: bytes, threads: int = 1) -> bytes:
    n = len(scalars_be32) // 32
    assert len(points_be96) >= 96 * n
    out = ctypes.create_string_buffer(48)
    _chk(lib().orc_msm(points_be96, scalars_be32, n, threads, out), "msm")
    return out.raw


class PreparedMsm:
    """Inputs decoded once so bench.py can time the CPU MSM alone."""

    def __init__(self, points_be96: bytes, scalars_be32: bytes):
        self.n = len(scalars_be32) // 32
        self._h = lib().orc_msm_prepare(points_be96, scalars_be32, self.n)
        if not self._h:
            raise ValueError("oracle msm_prepare failed")

    def run(self, threads: int = 1) -> bytes:
        out = ctypes.create_string_buffer(48)
        _chk(lib().orc_msm_run(self._h, threads, out), "msm_run")
        return out.raw

    def close(self) -> None:
        if self._h:
            lib().orc_msm_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def commit(slice_be96: bytes, row_be32: bytes, evaluation_form: bool = True, threads: int = 1) -> bytes:
    T = len(row_be32) // 32
    out = ctypes.create_string_buffer(48)
    _chk(lib().orc_commit(slice_be96, row_be32, T, int(evaluation_form), threads, out), "commit")
    return out.raw


def open_(slice_be96: bytes, row_be32: bytes, alpha_be32: bytes, evaluation_form: bool = True,
          threads: int = 1) -> Tuple[bytes, bytes]:
    T = len(row_be32) // 32
    ev = ctypes.create_string_buffer(32)
    pf = ctypes.create_string_buffer(48)
    _chk(lib().orc_open(slice_be96, row_be32, T, int(evaluation_form), alpha_be32, threads, ev, pf), "open")
    return ev.raw, pf.raw
