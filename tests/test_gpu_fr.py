"""GPU parity tests (`-m gpu`), fr: the Fr side: NTT (golden, oracle, 2^22, kernel variants), evaluation / opening kernels, the reference's Fr known-answer vector.
Every result of the HIP path, obtained through the C-ABI, is compared bit-for-bit with the CPU oracle on the same seeded inputs,
with the committed golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent properties (trapdoor
identity [f(tau)]G, linearity, NTT round trip).  All arithmetic is integer: the bar is bit-exact, no tolerance anywhere."""
import base64  # noqa: F401
import json  # noqa: F401
import os
import random  # noqa: F401

import numpy as np  # noqa: F401
import pytest

from oracle import bls12_381 as o  # noqa: F401
from oracle import cpu as oc  # noqa: F401
from tests.gpu_common import ROOT, H, ints, rand_scalars_bytes  # noqa: F401

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ NTT / eval
def test_ntt_golden_and_roundtrip(hip, golden_ntt):
    eng = hip()
    for case in golden_ntt:
        a = b"".join(H(v) for v in case["input"])
        assert eng.ntt(a, False) == b"".join(H(v) for v in case["forward"])
        assert eng.ntt(a, True) == b"".join(H(v) for v in case["inverse"])


@pytest.mark.parametrize("lg", [1, 3, 10, 11, 13, 16])
def test_ntt_matches_c_oracle(hip, lg):
    eng = hip()
    a = rand_scalars_bytes(1 << lg, 40 + lg)
    f = eng.ntt(a, False)
    assert f == oc.fr_ntt(a, False)
    assert eng.ntt(a, True) == oc.fr_ntt(a, True)
    assert eng.ntt(f, True) == a


def test_ntt_2_22_matches_c_oracle_directly_and_roundtrip(hip):
    """BASELINE configs[2] size: forward AND inverse 2^22-point transforms equal the C oracle's element for element (the
    API they serve: Client.fft, reference neurons/validator.py:59-65), round trip, X_0 = sum a_j; and the sizes around the
    radix-2 / register-blocked kernel switch (2^17 .. 2^19) plus both kernels forced on 2^18."""
    eng = hip()
    n = 1 << 22
    a_b = rand_scalars_bytes(n, 8)
    fa = eng.ntt(a_b, False)
    assert fa == oc.fr_ntt(a_b, False)                       # direct, all 4 M outputs
    assert eng.ntt(fa, True) == a_b
    ia = eng.ntt(a_b, True)
    assert ia == oc.fr_ntt(a_b, True)
    assert int.from_bytes(fa[:32], "big") == sum(ints(a_b)) % o.R
    for lg in (17, 18, 19):
        v = rand_scalars_bytes(1 << lg, 80 + lg)
        assert eng.ntt(v, False) == oc.fr_ntt(v, False) and eng.ntt(v, True) == oc.fr_ntt(v, True), lg


@pytest.mark.parametrize("env", [{"KZG_NTT_RADIX2": "1"}, {"KZG_NTT_TILE_LOG": "11"}, {"KZG_NTT_TILE_LOG": "9"}])
def test_ntt_kernel_variants_agree(env):
    """The A/B forms kept in the library (the radix-2 kernel forced at every size; the register-blocked kernel with 2048-
    and 512-element tiles) give the oracle's transforms too -- sizes on both sides of the kernel switch."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ntt_variant_check.py"), "18", "19", "20"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.mark.parametrize("env", [{"KZG_POLY_LDS_MIN_LOG": "18"}, 
                                 {"KZG_POLY_NO_LDS": "1"}])
def test_poly_kernel_variants_agree(env):
    """Opening kernels of long rows: the LDS-staged level-0 fold / quotient (default from 2^22 coefficients; forced from
    2^18 here) and the strided forms they replace (KZG_POLY_NO_LDS=1) both give the oracle's evaluation, quotient
    commitment and eval() -- alpha = random, 0, 1, a root of unity, r - 1."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "poly_variant_check.py"), "18", "19"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_eval_reference_kat_on_gpu(hip, fr_kat):
    """The reference's only arithmetic known-answer vector (tests/test_miner.py:33-55), through the HIP path."""
    from zkp_subnet_amd import codec

    eng = hip()
    y = eng.eval(codec.fr_list_to_be32(fr_kat["poly"]), codec.fr_to_be32(fr_kat["point"]))
    assert codec.be32_to_fr(y) == fr_kat["eval"]


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4096, 70001])
def test_eval_ragged_lengths(hip, n):
    eng = hip()
    c = rand_scalars_bytes(n, n)
    x = rand_scalars_bytes(1, n + 1)
    assert eng.eval(c, x) == oc.fr_eval(c, x)
    assert eng.eval(c, bytes(32)) == c[:32]                      # alpha = 0
