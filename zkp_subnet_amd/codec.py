"""Wire codec of the `Prove` synapse fields (reference base/protocol.py:24-60).

Fr  : 43-char unpadded std-alphabet base64 of 32 bytes big-endian  (pinned by the reference KAT,
      tests/test_miner.py:33-55: TEST_EVAL == Horner(TEST_POLY, TEST_POINT) only under this decoding)
G1  : 64-char base64 of the 48-byte ZCash compressed encoding (reference tests/test_validator.py:79-86 decodes
      proofs without re-padding, so the byte length is a multiple of 3)
Bulk polynomial decode goes through the native (host-side) codec when the library is built."""
from __future__ import annotations

import base64
import ctypes
from typing import List, Sequence

from . import _native

try:                                    # csrc/wire_py.c, built by zkp_subnet_amd.build
    from . import _wire
except ImportError:                     # not built yet: the C-ABI codec (kzg_b64_*_fr) below does the same job, slower
    _wire = None

R_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class CodecError(ValueError):
    pass


def fr_list_to_be32(poly: Sequence[str]) -> bytes:
    """List of 43-char strings -> n x 32 bytes big-endian.  Canonicity (< r) is enforced on the device."""
    n = len(poly)
    if n == 0:
        return b""
    if _wire is not None:               # no join, no intermediate copy, threaded decode with the GIL released
        try:
            return _wire.decode_fr_list(poly)
        except ValueError as e:
            raise CodecError(str(e)) from e
    if _native.lib_available():
        try:
            packed = "".join(poly).encode("ascii")
        except UnicodeEncodeError as e:
            raise CodecError("polynomial entries must be base64 text") from e
        if len(packed) != 43 * n:
            raise CodecError("every Fr must be 43 base64 characters (32 bytes, unpadded)")
        out = ctypes.create_string_buffer(32 * n)
        if _native.load().kzg_b64_decode_fr(packed, n, out) != 0:
            raise CodecError("invalid base64 in polynomial")
        return out.raw
    out = bytearray()
    for s in poly:
        out += fr_to_be32(s)
    return bytes(out)


def fr_to_be32(s: str) -> bytes:
    try:
        raw = base64.b64decode(s + "=" * (-len(s) % 4), validate=True)
    except Exception as e:
        raise CodecError("invalid base64 Fr") from e
    if len(raw) != 32:
        raise CodecError("Fr must decode to 32 bytes")
    return raw


def be32_to_fr(b: bytes) -> str:
    assert len(b) == 32
    return base64.b64encode(b).decode().rstrip("=")


def be32_to_fr_list(b: bytes) -> List[str]:
    n = len(b) // 32
    if n and _wire is not None:
        return _wire.encode_fr_list(b)
    if n and _native.lib_available():
        out = ctypes.create_string_buffer(43 * n)
        _native.load().kzg_b64_encode_fr(b, n, out)
        txt = out.raw.decode("ascii")
        return [txt[43 * i : 43 * i + 43] for i in range(n)]
    return [be32_to_fr(b[32 * i : 32 * i + 32]) for i in range(n)]


def g1_to_b64(c48: bytes) -> str:
    assert len(c48) == 48
    return base64.b64encode(c48).decode()


def g1_from_b64(s: str) -> bytes:
    try:
        raw = base64.b64decode(s, validate=True)
    except Exception as e:
        raise CodecError("invalid base64 G1") from e
    if len(raw) != 48:
        raise CodecError("G1 must decode to 48 bytes (ZCash compressed)")
    return raw
