"""Host-side opening verifier (pairing check) -- binding of the kzg_vk_* entry points.  No GPU involved: this is what
`Client.worker_verify` runs (reference neurons/validator.py:77-86)."""
from __future__ import annotations

import ctypes
from typing import Sequence

from . import _native
from ._native import KzgError


class Verifier:
    def __init__(self, handle):
        self._lib = _native.load()
        self._h = handle

    @classmethod
    def synthetic(cls, tau_x: int, factors: Sequence[int]) -> "Verifier":
        """Verifier key for a tau-derived SRS: [tau_x]_2 and [s0_k]_1 per resident slice (s0_k = L_i(tau_y))."""
        lib = _native.load()
        h = ctypes.c_void_p()
        s0 = b"".join(int(f).to_bytes(32, "big") for f in factors)
        rc = lib.kzg_vk_create_synthetic(int(tau_x).to_bytes(32, "big"), s0, len(s0) // 32, ctypes.byref(h))
        if rc != 0:
            raise KzgError(rc, "kzg_vk_create_synthetic failed")
        return cls(h)

    @classmethod
    def from_points(cls, tau_g2_be192: bytes, li_g1_be96: bytes) -> "Verifier":
        lib = _native.load()
        h = ctypes.c_void_p()
        rc = lib.kzg_vk_create(tau_g2_be192, li_g1_be96, len(li_g1_be96) // 96, ctypes.byref(h))
        if rc != 0:
            raise KzgError(rc, "kzg_vk_create failed: bad G2 / G1 key material")
        return cls(h)

    def export(self, n_slices: int) -> bytes:
        """192 B [tau_x]_2 followed by 96 B [L_i(tau_y)]_1 per slice: the contents of a `<setup>.vk` file."""
        out = ctypes.create_string_buffer(192 + 96 * n_slices)
        rc = self._lib.kzg_vk_export(self._h, out, len(out))
        if rc < 0:
            raise KzgError(rc, "kzg_vk_export failed")
        return out.raw[: 192 + 96 * rc]

    def verify(self, i: int, proof48: bytes, alpha32: bytes, eval32: bytes, commitment48: bytes) -> bool:
        if len(proof48) != 48 or len(commitment48) != 48:
            return False
        if len(alpha32) != 32 or len(eval32) != 32:          # the C side reads exactly 32 bytes of each
            raise KzgError(_native.KZG_E_ARG, "kzg_vk_verify: alpha / eval must be 32 bytes")
        ok = ctypes.c_int(0)
        rc = self._lib.kzg_vk_verify(self._h, i, proof48, alpha32, eval32, commitment48, ctypes.byref(ok))
        if rc != 0:
            raise KzgError(rc, "kzg_vk_verify: bad argument (index or non-canonical scalar)")
        return bool(ok.value)

    def verify_batch(self, indices: Sequence[int], proofs48: Sequence[bytes], alpha32: bytes, evals32: Sequence[bytes],
                     commitments48: Sequence[bytes], threads: int = 16) -> bool:
        """True only if EVERY (index, proof, eval, commitment) row verifies against the common alpha: one pairing check
        on a random linear combination of the rows (2^-128 soundness error).  False says nothing about which row."""
        n = len(indices)
        if not (n == len(proofs48) == len(evals32) == len(commitments48)):
            raise ValueError("verify_batch: ragged input")
        if len(alpha32) != 32 or any(len(e) != 32 for e in evals32):   # a short eval would have the C side read past
            raise KzgError(_native.KZG_E_ARG, "kzg_vk_verify_batch: alpha / evals must be 32 bytes each")   # the joined buffer
        if any(len(p) != 48 for p in proofs48) or any(len(c) != 48 for c in commitments48):
            return False
        idx = (ctypes.c_uint32 * max(n, 1))(*indices)
        ok = ctypes.c_int(0)
        rc = self._lib.kzg_vk_verify_batch(self._h, n, idx, b"".join(proofs48), alpha32, b"".join(evals32),
                                           b"".join(commitments48), threads, ctypes.byref(ok))
        if rc != 0:
            raise KzgError(rc, "kzg_vk_verify_batch: bad argument (index or non-canonical scalar)")
        return bool(ok.value)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.kzg_vk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
