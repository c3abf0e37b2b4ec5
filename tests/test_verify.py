"""CPU-only tests of the product's host-side pairing verifier (csrc/pairing_host.cpp) against the
independent pure-Python pairing of oracle/pairing.py and against the trapdoor check, on the golden KZG vectors.
Restates reference tests/test_miner.py:101-111 (verify(open, commit) is valid) and tests/test_validator.py:79-86,
103-104 (a proof bumped by one as a big-endian integer is rejected)."""
import ctypes
import random

import pytest

from oracle import bls12_381 as o
from oracle import pairing as pr
from zkp_subnet_amd import _native
from zkp_subnet_amd.build import build
from zkp_subnet_amd.verifier import Verifier

H = bytes.fromhex


@pytest.fixture(scope="module")
def lib():
    build()
    return _native.load()


def tower_to_poly(b):
    """12 x 48 B in tower order (w^h v^j u^k, k fastest) -> coefficients in Fp[w]/(w^12 - 2w^6 + 2), u = w^6 - 1."""
    c = [int.from_bytes(b[48 * k:48 * k + 48], "big") for k in range(12)]
    acc, idx = [0] * 12, 0
    u = [o.P - 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]
    for h in range(2):
        for j in range(3):
            for k in range(2):
                t = [c[idx]] + [0] * 11
                idx += 1
                if k:
                    t = pr.f12_mul(t, u)
                sh = [0] * 12
                sh[2 * j + h] = 1
                t = pr.f12_mul(t, sh)
                acc = [(x + y) % o.P for x, y in zip(acc, t)]
    return acc


def test_pairing_matches_independent_python_pairing(lib):
    out = ctypes.create_string_buffer(576)
    assert lib.kzg_vk_pairing(o.g1_to_be96(o.G1), pr.g2_to_be192(pr.G2), out) == 0
    e11 = pr.f12_pow(pr.miller_loop(o.G1, pr.G2), pr.FINAL_EXP)
    assert tower_to_poly(out.raw) == e11
    assert e11 != pr.F12_ONE and pr.f12_pow(e11, o.R) == pr.F12_ONE
    a, b = 0x1234567, 0x7654321
    assert lib.kzg_vk_pairing(o.g1_to_be96(o.g1_mul(o.G1, a)), pr.g2_to_be192(pr.g2_mul(pr.G2, b)), out) == 0
    assert tower_to_poly(out.raw) == pr.f12_pow(e11, a * b)          # bilinearity, exact Fp12 value
    # off-curve inputs are refused
    bad = bytearray(o.g1_to_be96(o.G1))
    bad[-1] ^= 1
    assert lib.kzg_vk_pairing(bytes(bad), pr.g2_to_be192(pr.G2), out) == _native.KZG_E_POINT


def test_verify_golden_vectors_and_rejections(golden_kzg):
    tx, ty = int(golden_kzg["tau_x"], 16), int(golden_kzg["tau_y"], 16)
    for case in golden_kzg["cases"]:
        ms, i = case["machines_scale"], case["i"]
        li = o.lagrange_at(i, 1 << ms, ty)
        vk = Verifier.synthetic(tx, [li])
        proof, alpha, ev, com = H(case["proof"]), H(case["alpha"]), H(case["eval"]), H(case["commitment"])
        assert vk.verify(0, proof, alpha, ev, com), case["name"]
        wrong_eval = ((int(case["eval"], 16) + 1) % o.R).to_bytes(32, "big")
        assert not vk.verify(0, proof, alpha, wrong_eval, com), case["name"]
        bumped = (int.from_bytes(proof, "big") + 1).to_bytes(48, "big")        # reference test_validator.py:79-86
        assert not vk.verify(0, bumped, alpha, ev, com), case["name"]
        if com != proof:                                                      # (zero polynomial: both are infinity)
            assert not vk.verify(0, com, alpha, ev, proof), case["name"]      # swapped
        vk.close()


def test_verify_agrees_with_python_pairing_and_trapdoor():
    rnd = random.Random(4)
    tx, ty = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
    ms, scale = 2, 5
    srs = [o.srs_slice(tx, ty, scale, ms, i) for i in range(4)]
    lis = [o.lagrange_at(i, 4, ty) for i in range(4)]
    vk = Verifier.synthetic(tx, lis)
    tau_g2 = pr.g2_mul(pr.G2, tx)
    vk2 = Verifier.from_points(pr.g2_to_be192(tau_g2), b"".join(o.g1_to_be96(o.g1_table().mul(li)) for li in lis))
    for i in (0, 3):
        row = [rnd.randrange(o.R) for _ in range(8)]
        alpha = rnd.randrange(o.R)
        c = o.worker_commit(srs[i], row)
        y, pi = o.worker_open(srs[i], row, alpha)
        args = (o.g1_compress(pi), alpha.to_bytes(32, "big"), y.to_bytes(32, "big"), o.g1_compress(c))
        assert vk.verify(i, *args) and vk2.verify(i, *args)
        assert pr.kzg_verify(c, pi, alpha, y, o.g1_table().mul(lis[i]), tau_g2)
        assert o.verify_trapdoor(tx, ty, ms, i, c, pi, alpha, y)
        assert not vk.verify((i + 1) % 4, *args)                              # another worker's basis
        assert not pr.kzg_verify(c, pi, alpha, (y + 5) % o.R, o.g1_table().mul(lis[i]), tau_g2)
    # points of the curve outside the r-torsion subgroup must not verify: x = 4 gives a point of E(Fp) with cofactor
    x = 4
    while o.fp_sqrt((x ** 3 + 4) % o.P) is None:
        x += 1
    stray = (x, o.fp_sqrt((x ** 3 + 4) % o.P))
    assert o.g1_add(o.g1_mul(stray, o.R - 1), stray) is not None          # [r]P != infinity: not in G1
    assert not vk.verify(0, o.g1_compress(stray), (1).to_bytes(32, "big"), (1).to_bytes(32, "big"), o.g1_compress(o.G1))
    with pytest.raises(Exception):
        vk.verify(9, *args)                                                   # slice index out of range
    with pytest.raises(Exception):
        vk.verify(0, args[0], o.R.to_bytes(32, "big"), args[2], args[3])      # non-canonical alpha
    assert not vk.verify(0, b"\x00" * 48, args[1], args[2], args[3])          # not a compressed point


def test_vk_export_roundtrip():
    rnd = random.Random(8)
    tx, lis = rnd.randrange(1, o.R), [rnd.randrange(1, o.R) for _ in range(3)]
    vk = Verifier.synthetic(tx, lis)
    blob = vk.export(3)
    assert blob[:192] == pr.g2_to_be192(pr.g2_mul(pr.G2, tx))
    assert blob[192:] == b"".join(o.g1_to_be96(o.g1_table().mul(li)) for li in lis)
    vk2 = Verifier.from_points(blob[:192], blob[192:])
    assert vk2.export(3) == blob


def test_bad_key_material_is_refused():
    from zkp_subnet_amd import KzgError

    good_g2 = pr.g2_to_be192(pr.G2)
    bad = bytearray(good_g2)
    bad[-1] ^= 1
    with pytest.raises(KzgError):
        Verifier.from_points(bytes(bad), o.g1_to_be96(o.G1))


def test_batch_verification_of_a_step_one_pairing_check():
    """kzg_vk_verify_batch: all rows of a step (common alpha) folded into ONE pairing check by random 128-bit weights.
    True exactly when every row is valid; a single wrong eval, a proof of another row, a row verified against another
    worker's basis, a point outside G1 or a malformed encoding make it False; thread counts do not matter."""
    rnd = random.Random(44)
    tx, ty = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
    ms, scale = 2, 5
    lis = [o.lagrange_at(i, 4, ty) for i in range(4)]
    vk = Verifier.synthetic(tx, lis)
    alpha = rnd.randrange(o.R)
    ab = alpha.to_bytes(32, "big")
    idx, proofs, evals, comms = [], [], [], []
    for i in (0, 1, 2, 3, 1, 3):
        srs = o.srs_slice(tx, ty, scale, ms, i)
        row = [rnd.randrange(o.R) for _ in range(8)]
        c = o.worker_commit(srs, row)
        y, pi = o.worker_open(srs, row, alpha)
        idx.append(i); proofs.append(o.g1_compress(pi)); evals.append(y.to_bytes(32, "big")); comms.append(o.g1_compress(c))
    for th in (1, 3, 16):
        assert vk.verify_batch(idx, proofs, ab, evals, comms, threads=th)
    assert vk.verify_batch([], [], ab, [], [])                                     # nothing to refute
    assert vk.verify_batch(idx[:1], proofs[:1], ab, evals[:1], comms[:1])
    bad_eval = list(evals)
    bad_eval[4] = ((int.from_bytes(evals[4], "big") + 1) % o.R).to_bytes(32, "big")
    assert not vk.verify_batch(idx, proofs, ab, bad_eval, comms)
    swapped = list(proofs)
    swapped[0], swapped[5] = swapped[5], swapped[0]
    assert not vk.verify_batch(idx, swapped, ab, evals, comms)
    assert not vk.verify_batch([idx[1]] + idx[1:], proofs, ab, evals, comms)       # row 0 against worker 1's basis
    assert not vk.verify_batch(idx, proofs, ((alpha + 1) % o.R).to_bytes(32, "big"), evals, comms)
    x = 4
    while o.fp_sqrt((x ** 3 + 4) % o.P) is None:
        x += 1
    stray = o.g1_compress((x, o.fp_sqrt((x ** 3 + 4) % o.P)))                      # on the curve, outside G1
    assert not vk.verify_batch(idx, [stray] + proofs[1:], ab, evals, comms)
    assert not vk.verify_batch(idx, [b"\x00" * 48] + proofs[1:], ab, evals, comms)  # not a compressed point
    with pytest.raises(Exception):
        vk.verify_batch([9] + idx[1:], proofs, ab, evals, comms)                   # slice index out of range
    with pytest.raises(Exception):
        vk.verify_batch(idx, proofs, o.R.to_bytes(32, "big"), evals, comms)        # non-canonical alpha
    # every row agrees with the row-by-row verifier
    assert all(vk.verify(i, p, ab, e, c) for i, p, e, c in zip(idx, proofs, evals, comms))
    vk.close()
