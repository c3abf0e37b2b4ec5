"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/kzg_mi355x.h
declares, refuses to run without a gfx950 device (no CPU fallback), and its host-side wire codec is exact."""
import base64
import ctypes
import os
import re

import pytest

from zkp_subnet_amd import _native, codec
from zkp_subnet_amd.build import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()  # hipcc cross-compiles gfx950 without a GPU
    return _native.load()


def test_header_symbols_all_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "kzg_mi355x.h")).read()
    declared = set(re.findall(r"\b(kzg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"kzg_ctx", "kzg_status"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _native.SYMBOLS, f"{name} has no ctypes prototype"
    assert set(_native.SYMBOLS) <= declared


def test_version(lib):
    assert b"gfx950" in lib.kzg_version()


def test_no_cpu_fallback_without_device(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert lib.kzg_create(0, ctypes.byref(h)) == _native.KZG_E_HIP
    from zkp_subnet_amd import HipEngine, KzgError

    with pytest.raises(KzgError):
        HipEngine(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "zkp_subnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hip.h", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libkzg_oracle" not in src and "kzg_cpu" not in src, f


def test_b64_codec_matches_python(lib, fr_kat):
    import random

    rnd = random.Random(9)
    vals = [rnd.randrange(1 << 256).to_bytes(32, "big") for _ in range(257)] + [bytes(32), b"\xff" * 32]
    raw = b"".join(vals)
    strs = codec.be32_to_fr_list(raw)
    assert strs == [base64.b64encode(v).decode().rstrip("=") for v in vals]
    assert all(len(s) == 43 for s in strs)
    assert codec.fr_list_to_be32(strs) == raw
    # the reference's own strings decode to the reference's own evaluation (KAT restated through the codec)
    from oracle import bls12_381 as o

    poly = o.fr_from_be32(codec.fr_list_to_be32(fr_kat["poly"]))
    assert o.poly_eval(poly, o.fr_from_b64(fr_kat["point"])) == o.fr_from_b64(fr_kat["eval"])


def test_b64_codec_rejects_garbage(lib):
    with pytest.raises(codec.CodecError):
        codec.fr_list_to_be32(["A" * 42])
    with pytest.raises(codec.CodecError):
        codec.fr_list_to_be32(["!" + "A" * 42])
    with pytest.raises(codec.CodecError):
        codec.fr_list_to_be32(["A" * 42 + "B"])  # non-zero padding bits: not the encoding of 32 bytes
    with pytest.raises(codec.CodecError):
        codec.g1_from_b64("AAAA")


def test_wire_extension_matches_python_base64():
    """csrc/wire_py.c (the List[str] fast path of the synapse codec): same bytes as Python's base64, same rejections."""
    import base64
    import os as _os

    from zkp_subnet_amd import codec
    from zkp_subnet_amd.build import build_wire

    build_wire()
    import importlib

    importlib.reload(codec)
    assert codec._wire is not None
    for n in (0, 1, 7, 3000, 40000):                   # 3000 / 40000: 4 / 8 pool threads
        raw = _os.urandom(32 * n)
        lst = codec.be32_to_fr_list(raw)
        assert lst == [base64.b64encode(raw[32 * i:32 * i + 32]).decode().rstrip("=") for i in range(n)]
        assert codec.fr_list_to_be32(lst) == raw
        assert codec.fr_list_to_be32(tuple(lst)) == raw
        if n:
            assert codec._wire.decode_fr_list(lst, 3) == raw
    good = codec.be32_to_fr(bytes(range(32)))
    for bad in ([good[:-1]], [good + "A"], [good[:-1] + "B"], ["\u00e9" * 43], [good.encode()], [good, 5], [good[:20] + "=" + good[21:]],
                [good] * 20000 + [good[:5] + "*" + good[6:]] + [good] * 20000):
        with pytest.raises(codec.CodecError):
            codec.fr_list_to_be32(bad)
    with pytest.raises(ValueError):
        codec._wire.encode_fr_list(b"\x00" * 33)
