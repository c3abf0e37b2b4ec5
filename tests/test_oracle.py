"""Pins the oracle: reference Fr KAT, public constants, Python-vs-C agreement on tests/golden/.

CPU only (`-m "not gpu"`).  The oracle is the checker for every GPU parity test, so it is pinned
first against the only known-answer data the reference holds (tests/test_miner.py:33-55).
"""
import random

import pytest

from oracle import bls12_381 as o

H = bytes.fromhex


def test_reference_fr_kat_python(fr_kat):
    poly = [o.fr_from_b64(s) for s in fr_kat["poly"]]
    assert len(poly) == 16
    assert o.poly_eval(poly, o.fr_from_b64(fr_kat["point"])) == o.fr_from_b64(fr_kat["eval"])
    # round trip of the wire encoding: 43 chars, unpadded
    for s in fr_kat["poly"]:
        assert len(s) == 43 and o.fr_to_b64(o.fr_from_b64(s)) == s


def test_reference_fr_kat_c(fr_kat, oracle_cpu):
    coeffs = o.fr_to_be32([o.fr_from_b64(s) for s in fr_kat["poly"]])
    x = o.fr_from_b64(fr_kat["point"]).to_bytes(32, "big")
    assert oracle_cpu.fr_eval(coeffs, x) == o.fr_from_b64(fr_kat["eval"]).to_bytes(32, "big")


# Public literals, hard-coded here (NOT read from tests/golden/constants.json, which gen_golden.py writes from this same
# oracle): the ZCash-compressed BLS12-381 G1 generator and the points 2G / -G (2G is the widely published BLS public key
# of secret key 2), the identity encoding, and the roots of unity SURVEY.md Appendix A derived independently of oracle/.
G1_C48 = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
TWO_G1_C48 = "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e"
NEG_G1_C48 = "b7f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
ROOTS = {
    12: 0x564C0A11A0F704F4FC3E8ACFE0F8245F0AD1347B378FBF96E206DA11A5D36306,
    20: 0x03E1C54BCB947035A57A6E07CB98DE4A2F69E02D265E09D9FECE7E0E39898D4B,
    22: 0x0ABE6A5E5ABCAA32F2D38F10FBB8D1BBE08FEC7C86389BEEC6E7A6FFB08E3363,
    32: 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B,
}


def test_constants(golden_constants):
    x = o.BLS_X
    assert o.R == x**4 - x**2 + 1
    assert o.P == (x - 1) ** 2 * o.R // 3 + x
    assert o.P == 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    assert o.R == 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    assert o.is_on_curve(o.G1) and o.G1[1] == G1_Y
    assert o.g1_mul(o.G1, o.R - 1) == o.g1_neg(o.G1)  # group order
    assert o.g1_compress(o.G1).hex() == G1_C48
    assert o.g1_compress(o.g1_mul(o.G1, 2)).hex() == TWO_G1_C48
    assert o.g1_compress(o.g1_neg(o.G1)).hex() == NEG_G1_C48
    assert o.g1_decompress(H(TWO_G1_C48)) == o.g1_add(o.G1, o.G1)
    assert o.g1_decompress(H(NEG_G1_C48)) == o.g1_neg(o.G1)
    assert o.g1_compress(o.g1_mul(o.G1, o.R)).hex() == "c0" + "00" * 47      # identity
    for k, w in ROOTS.items():
        assert o.root_of_unity(1 << k) == w
        assert pow(w, 1 << k, o.R) == 1 and pow(w, 1 << (k - 1), o.R) == o.R - 1
    # the committed fixture file agrees with the literals above (it is generated from the oracle: no extra evidence)
    assert golden_constants["g1_compressed"] == G1_C48
    assert golden_constants["two_g1_compressed"] == TWO_G1_C48
    assert golden_constants["neg_g1_compressed"] == NEG_G1_C48
    assert golden_constants["identity_compressed"] == "c0" + "00" * 47
    for k, v in golden_constants["roots_of_unity"].items():
        w = int(v, 16)
        assert pow(w, 1 << int(k), o.R) == 1 and pow(w, 1 << (int(k) - 1), o.R) == o.R - 1


def test_constants_c_oracle(oracle_cpu):
    """The same public literals through the C restatement (oracle/kzg_cpu.c): [1]G, [2]G, [r-1]G as one-point MSMs."""
    g = o.g1_to_be96(o.G1)
    for k, want in ((1, G1_C48), (2, TWO_G1_C48), (o.R - 1, NEG_G1_C48), (0, "c0" + "00" * 47)):
        assert oracle_cpu.msm(g, k.to_bytes(32, "big")).hex() == want


def test_non_canonical_fr_rejected():
    with pytest.raises(ValueError):
        o.fr_from_b64(o.base64.b64encode(o.R.to_bytes(32, "big")).decode().rstrip("="))
    with pytest.raises(ValueError):
        o.fr_from_b64("AAAA")


def test_compress_roundtrip():
    rnd = random.Random(5)
    for _ in range(8):
        pt = o.g1_table().mul(rnd.randrange(1, o.R))
        assert o.is_on_curve(pt)
        assert o.g1_decompress(o.g1_compress(pt)) == pt
        assert o.g1_from_be96(o.g1_to_be96(pt)) == pt
    assert o.g1_decompress(o.g1_compress(None)) is None


def test_ntt_golden_python_and_c(golden_ntt, oracle_cpu):
    for case in golden_ntt:
        a = [int(v, 16) for v in case["input"]]
        fwd = [int(v, 16) for v in case["forward"]]
        inv = [int(v, 16) for v in case["inverse"]]
        if case["n"] <= 16:
            assert o.dft_naive(a) == fwd and o.dft_naive(a, inverse=True) == inv
        assert o.ntt(a) == fwd and o.ntt(fwd, inverse=True) == a
        assert oracle_cpu.fr_ntt(o.fr_to_be32(a), False) == o.fr_to_be32(fwd)
        assert oracle_cpu.fr_ntt(o.fr_to_be32(a), True) == o.fr_to_be32(inv)


def test_msm_golden_c(golden_msm, oracle_cpu):
    for case in golden_msm:
        pts = b"".join(H(p) for p in case["points"])
        sc = b"".join(H(s) for s in case["scalars"])
        for threads in (1, 3):
            assert oracle_cpu.msm(pts, sc, threads).hex() == case["result"], case["name"]


def test_msm_golden_python_pippenger(golden_msm):
    for case in golden_msm:
        pts = [o.g1_from_be96(H(p)) for p in case["points"]]
        sc = [int(s, 16) for s in case["scalars"]]
        assert o.g1_compress(o.msm_pippenger(pts, sc, c=5)).hex() == case["result"], case["name"]


def test_kzg_golden_c(golden_kzg, oracle_cpu):
    tx, ty = H(golden_kzg["tau_x"]), H(golden_kzg["tau_y"])
    for case in golden_kzg["cases"]:
        srs = oracle_cpu.srs_gen(tx, ty, case["scale"], case["machines_scale"], case["i"])
        assert srs[:96].hex() == case["srs_first"] and srs[-96:].hex() == case["srs_last"], case["name"]
        row = b"".join(H(v) for v in case["row"])
        ef = case["evaluation_form"]
        assert oracle_cpu.commit(srs, row, ef).hex() == case["commitment"], case["name"]
        ev, pf = oracle_cpu.open_(srs, row, H(case["alpha"]), ef, threads=2)
        assert ev.hex() == case["eval"] and pf.hex() == case["proof"], case["name"]


def test_kzg_golden_python_small(golden_kzg):
    tx, ty = int(golden_kzg["tau_x"], 16), int(golden_kzg["tau_y"], 16)
    for case in golden_kzg["cases"]:
        if len(case["row"]) > 16:
            continue
        srs = o.srs_slice(tx, ty, case["scale"], case["machines_scale"], case["i"])
        row = [int(v, 16) for v in case["row"]]
        c = o.worker_commit(srs, row, case["evaluation_form"])
        y, pi = o.worker_open(srs, row, int(case["alpha"], 16), case["evaluation_form"])
        assert o.g1_compress(c).hex() == case["commitment"]
        assert y == int(case["eval"], 16) and o.g1_compress(pi).hex() == case["proof"]


def test_structural_properties_P1_P4(golden_kzg):
    """Reference test structure restated with the trapdoor check:
    open verifies against commit; eval == eval(IFFT_left(row), alpha); a proof whose big-endian
    integer is incremented by one is rejected (reference tests/test_validator.py:79-86)."""
    tx, ty = int(golden_kzg["tau_x"], 16), int(golden_kzg["tau_y"], 16)
    case = next(c for c in golden_kzg["cases"] if c["name"] == "T16_tests_shape")
    ms, i = case["machines_scale"], case["i"]
    row = [int(v, 16) for v in case["row"]]
    alpha, y = int(case["alpha"], 16), int(case["eval"], 16)
    c = o.g1_decompress(H(case["commitment"]))
    pi = o.g1_decompress(H(case["proof"]))
    assert y == o.poly_eval(o.ntt(row, inverse=True), alpha)  # P3
    assert o.verify_trapdoor(tx, ty, ms, i, c, pi, alpha, y)  # P1
    bumped = (int.from_bytes(H(case["proof"]), "big") + 1).to_bytes(48, "big")  # P4
    try:
        bad = o.g1_decompress(bumped)
    except AssertionError:
        bad = None  # not even a curve point -> rejected
    assert bad is None or not o.verify_trapdoor(tx, ty, ms, i, c, bad, alpha, y)


def test_c_oracle_trapdoor_midsize(oracle_cpu):
    """2^12 (config 1 size): C Pippenger == [f(tau)]G computed through an independent route."""
    rnd = random.Random(11)
    tx, ty = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
    srs = oracle_cpu.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), 12, 0, 0)
    coeffs = [rnd.randrange(o.R) for _ in range(1 << 12)]
    got = oracle_cpu.commit(srs, o.fr_to_be32(coeffs), evaluation_form=False, threads=4)
    assert got == o.g1_compress(o.trapdoor_commit(tx, ty, 0, 0, coeffs))


def test_endomorphism_subgroup_criterion_constants():
    """The G1 membership test of csrc/g1_kernels.hip (k_g1_subgroup_check_lp): sigma(x, y) = (beta x, y) acts on G1 as -z^2,
    and z^4 - z^2 + 1 = r is the degree of sigma + z^2, so [z^2]P == -sigma(P) holds exactly on G1.  Checks the
    constants the kernel hard-codes, on the generator and on an on-curve point outside the subgroup."""
    z = abs(o.BLS_X)
    assert z == 0xD201000000010000 and z ** 4 - z ** 2 + 1 == o.R
    beta = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE
    assert beta != 1 and pow(beta, 3, o.P) == 1
    src = open(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "zkp_subnet_amd", "csrc", "g1_kernels.hip")).read()
    bm = beta * pow(2, 392, o.P) % o.P            # Montgomery residue, 14 limbs of 28 bits, as the kernel stores it
    for i in range(14):
        assert "0x%08xu" % ((bm >> (28 * i)) & 0xFFFFFFF) in src

    def mul(pt, k):                               # no reduction of k mod r: the point may lie outside G1
        acc, base = o.JAC_INF, o.to_jac(pt)
        while k:
            if k & 1:
                acc = o.jac_add(acc, base)
            base = o.jac_double(base)
            k >>= 1
        return o.to_affine(acc)

    g = o.G1
    q = o.g1_neg(mul(g, z * z))
    assert (beta * g[0] % o.P, g[1]) == (q[0], q[1])
    x = 5                                          # (5, sqrt(129)) is on the curve, order not dividing r
    y = o.fp_sqrt((x ** 3 + 4) % o.P)
    assert y is not None and o.is_on_curve((x, y)) and mul((x, y), o.R) is not None
    q = o.g1_neg(mul((x, y), z * z))
    assert (beta * x % o.P, y) != (q[0], q[1])


def test_c_oracle_thread_counts_are_capped_and_agree(oracle_cpu):
    """The C oracle spreads (window x chunk) tasks over a thread pool; any thread count -- also far above the CPUs this
    process may use (the count is capped at the affinity mask / cgroup quota) -- gives the same point, for uniform and for
    skewed scalars (the latter exercise the batched-affine collision / spill path)."""
    assert oracle_cpu.usable_cpus() >= 1
    rnd = random.Random(31)
    n = 3000
    srs = oracle_cpu.srs_gen((9).to_bytes(32, "big"), (1).to_bytes(32, "big"), 12, 0, 0)[: 96 * n]
    for kind in ("uniform", "equal", "few", "tiny"):
        if kind == "uniform":
            sc = [rnd.randrange(o.R) for _ in range(n)]
        elif kind == "equal":
            sc = [rnd.randrange(o.R)] * n
        elif kind == "few":
            pool = [rnd.randrange(o.R) for _ in range(3)]
            sc = [rnd.choice(pool) for _ in range(n)]
        else:
            sc = [rnd.randrange(4) for _ in range(n)]
        sb = o.fr_to_be32(sc)
        ref = oracle_cpu.msm(srs, sb, threads=1)
        for th in (2, 7, 64, 100000):
            assert oracle_cpu.msm(srs, sb, threads=th) == ref, (kind, th)
    pts = [o.g1_from_be96(srs[96 * k:96 * k + 96]) for k in range(200)]
    sc = [rnd.randrange(o.R) for _ in range(200)]
    assert oracle_cpu.msm(srs[: 96 * 200], o.fr_to_be32(sc), threads=3) == o.g1_compress(o.msm_pippenger(pts, sc))
