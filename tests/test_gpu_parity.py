"""GPU parity tests proper (`-m gpu`): every result of the HIP path, obtained through the C-ABI, is compared
bit-for-bit with the CPU oracle on the same seeded inputs, with the committed golden fixtures, and -- at
BASELINE.json's full sizes -- through size-independent properties (trapdoor identity [f(tau)]G, linearity,
NTT round trip).  All arithmetic is integer: the bar is bit-exact, no tolerance anywhere."""
import base64
import json
import os
import random

import numpy as np
import pytest

from oracle import bls12_381 as o
from oracle import cpu as oc

pytestmark = pytest.mark.gpu
H = bytes.fromhex
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    from zkp_subnet_amd import HipEngine

    engines = []

    def make(window=0):
        e = HipEngine(0, window=window)   # raises if the HIP library is missing or no gfx950 device works
        engines.append(e)
        return e

    yield make
    for e in engines:
        e.close()


def rand_scalars_bytes(n, seed):
    raw = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw[:, 0] &= 0x3F                       # < 2^254 < r: canonical
    return raw.tobytes()


def ints(b):
    return [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]


# ------------------------------------------------------------------ unit ops
@pytest.mark.parametrize("field,mod,w", [(0, o.P, 48), (1, o.R, 32)])
def test_field_ops_bit_exact(hip, field, mod, w):
    eng = hip()
    rnd = random.Random(100 + field)
    edge = [0, 1, 2, mod - 1, mod - 2, (mod + 1) // 2, (1 << (8 * w - 3)) % mod, 0xFFFFFFFF, (1 << 32), mod >> 1]
    va = [rnd.randrange(mod) for _ in range(100000)] + [x for x in edge for _ in edge]
    vb = [rnd.randrange(mod) for _ in range(100000)] + [y for _ in edge for y in edge]
    a = b"".join(v.to_bytes(w, "big") for v in va)
    b = b"".join(v.to_bytes(w, "big") for v in vb)
    ops = [(0, lambda x, y: x * y % mod), (1, lambda x, y: (x + y) % mod), (2, lambda x, y: (x - y) % mod),
           (3, lambda x, y: x * y % mod), (4, lambda x, y: x * x % mod)]
    if field == 1:   # the product-free reductions of fr29.hip.h on lazily accumulated sums of up to 58 r
        lazy = lambda x, y: (1 + (x & (2**64 - 1)) % 29) * (x + y) % mod   # noqa: E731
        ops += [(5, lazy), (6, lazy)]
    for op, fn in ops:
        out = eng.test_field(field, op, a, b)
        exp = b"".join(fn(x, y).to_bytes(w, "big") for x, y in zip(va, vb))
        assert out == exp, f"field {field} op {op}"


def test_g1_ops_bit_exact_including_exceptional_cases(hip):
    eng = hip()
    rnd = random.Random(5)
    tb = o.g1_table()
    pa = [tb.mul(rnd.randrange(1, o.R)) for _ in range(200)]
    pb = [tb.mul(rnd.randrange(1, o.R)) for _ in range(200)]
    pb[0] = pa[0]                 # P + P inside the mixed add
    pb[1] = o.g1_neg(pa[1])       # P + (-P) = infinity
    pb[2] = None
    pa[3] = None
    pa[4] = pb[4] = None
    pb[5] = o.g1_add(pa[5], pa[5])            # 2a + b with b == 2a: equal points inside a FULL addition (ops 1, 5)
    pb[6] = o.g1_neg(o.g1_add(pa[6], pa[6]))  # 2a + (-2a) = infinity
    a = b"".join(o.g1_to_be96(p) for p in pa)
    b = b"".join(o.g1_to_be96(p) for p in pb)
    def chain(x, y):                      # a, then 40 alternating mixed adds of b, a, b, a ...
        r = x
        for k in range(40):
            r = o.g1_add(r, x if k & 1 else y)
        return r

    def chain_lp(x, y):                   # ten rounds r <- 2r + b from a (lane-parallel doubling / addition)
        r = x
        for _ in range(10):
            r = o.g1_add(o.g1_add(r, r), y)
        return r

    exp = {0: lambda x, y: o.g1_add(x, y), 1: lambda x, y: o.g1_add(o.g1_add(x, x), y),
           2: lambda x, y: o.g1_add(x, x), 3: lambda x, y: o.g1_mul(x, 4) if x else None, 4: chain,
           5: lambda x, y: o.g1_add(o.g1_add(x, x), y), 6: lambda x, y: o.g1_mul(x, 4) if x else None, 7: chain_lp}
    for op in range(8):
        out = eng.test_g1(op, a, b)
        assert out == b"".join(o.g1_to_be96(exp[op](x, y)) for x, y in zip(pa, pb)), f"g1 op {op}"


# ------------------------------------------------------------------ SRS + window tables
@pytest.mark.parametrize("window", [0, 4, 7, 13])
def test_srs_generation_and_window_tables(hip, window):
    eng = hip(window)
    tx, ty = 0x1234567 + window, 0xABCDEF1
    eng.gen_srs(tx, ty, 6, 2)
    exp = b"".join(oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), 6, 2, i) for i in range(4))
    assert eng.srs_read(0, 64) == exp
    offs = eng.window_offsets
    assert offs[0] == 0 and offs[-1] == 256 and max(b - a for a, b in zip(offs, offs[1:])) == eng.window
    for w in (1, len(offs) // 2, len(offs) - 2):
        tab = eng.srs_read(5, 3, window=w)
        for j in range(3):
            pt = o.g1_from_be96(exp[96 * (5 + j):96 * (6 + j)])
            assert tab[96 * j:96 * j + 96] == o.g1_to_be96(o.g1_mul(pt, pow(2, offs[w], o.R)))


def test_load_srs_roundtrip_and_rejects_bad_points(hip):
    from zkp_subnet_amd import KzgError

    eng = hip()
    srs = oc.srs_gen((77).to_bytes(32, "big"), (1).to_bytes(32, "big"), 4, 0, 0)
    eng.load_srs(srs, 4, 0)
    assert eng.srs_read(0, 16) == srs
    bad = bytearray(srs)
    bad[95] ^= 1                                              # y off the curve
    with pytest.raises(KzgError):
        hip().load_srs(bytes(bad), 4, 0)
    # ON the curve but outside G1 (x = 5; E(Fp) has a ~2^126 cofactor): refused by the membership test of the loaders,
    # uncompressed and compressed; accepted only when that test is switched off explicitly
    y5 = o.fp_sqrt((5 ** 3 + 4) % o.P)
    rogue = srs[:96 * 7] + o.g1_to_be96((5, y5)) + srs[96 * 8:]
    for data, comp in ((rogue, False), (b"".join(o.g1_compress(o.g1_from_be96(rogue[96 * k:96 * k + 96])) for k in range(16)), True)):
        e2 = hip()
        with pytest.raises(KzgError) as ei:
            e2.load_srs(data, 4, 0, compressed=comp)
        assert "subgroup" in str(ei.value)
        e2.set_srs_subgroup_check(False)
        e2.load_srs(data, 4, 0, compressed=comp)
        assert e2.srs_read(7, 1) == o.g1_to_be96((5, y5))
        with pytest.raises(KzgError):                          # the opt-out covered ONE load: the check is armed again
            e2.load_srs(data, 4, 0, compressed=comp)
        assert e2.srs_read(7, 1) == o.g1_to_be96((5, y5))      # ... and the refused load left the installed SRS serving
    with pytest.raises(KzgError):
        hip().load_srs(o.P.to_bytes(48, "big") * 2 + srs[96:], 4, 0)   # unreduced coordinate


# ------------------------------------------------------------------ MSM
def test_msm_golden_edge_cases(hip, golden_msm):
    for window in (5, 9):
        for case in golden_msm:
            n = len(case["points"])
            if n == 0:
                continue
            npad = 1 << max(0, (n - 1).bit_length())
            eng = hip(window)
            eng.load_srs(b"".join(H(p) for p in case["points"]) + bytes(96 * (npad - n)), npad.bit_length() - 1, 0)
            got = eng.msm(b"".join(H(s) for s in case["scalars"]), 0)
            assert got.hex() == case["result"], (case["name"], window)
            eng.close()


def test_msm_empty_is_infinity(hip):
    eng = hip()
    eng.gen_srs(3, 1, 4, 0)
    assert eng.msm(b"", 0) == bytes([0xC0]) + bytes(47)


def test_msm_rejects_non_canonical_scalar(hip):
    from zkp_subnet_amd import KzgError

    eng = hip()
    eng.gen_srs(3, 1, 4, 0)
    with pytest.raises(KzgError) as ei:
        eng.msm(o.R.to_bytes(32, "big") + bytes(32), 0)
    assert ei.value.code == -2
    with pytest.raises(KzgError):
        eng.msm(bytes(32) * 17, 0)                            # longer than the resident SRS


@pytest.mark.parametrize("lg,window", [(4, 0), (9, 0), (10, 6), (12, 0), (13, 11), (14, 0)])
def test_msm_matches_c_oracle(hip, lg, window):
    eng = hip(window)
    tx = 0xC0FFEE + lg
    eng.gen_srs(tx, 1, lg, 0)
    n = 1 << lg
    sc = rand_scalars_bytes(n, lg)
    srs = eng.srs_read(0, n)
    assert srs == oc.srs_gen(tx.to_bytes(32, "big"), (1).to_bytes(32, "big"), lg, 0, 0)
    got = eng.msm(sc, 0)
    assert got == oc.msm(srs, sc, threads=8)
    assert got == oc.g1_mul_gen(o.poly_eval(ints(sc), tx).to_bytes(32, "big"))        # trapdoor route
    # ragged length and an offset window into the SRS
    m, off = n - 3, 2
    assert eng.msm(sc[: 32 * m], off) == oc.msm(srs[96 * off:96 * (off + m)], sc[: 32 * m], threads=8)


def test_widest_window_24_msm_and_batched_commit_open(hip):
    """c = 24 (what 2^26-point slices get: 11 windows, 2^23 buckets) on a small input, where it is cheap to check against
    the oracle: the plain MSM, and a batched commit+open, whose sort key carries one more bit (24 in all: the widest the
    sort's 12 + 12 split takes)."""
    lg, n = 10, 1 << 10
    eng = hip(24)
    tx = 0x24C0DE
    eng.gen_srs(tx, 1, lg, 0)
    assert eng.window == 24
    sc = rand_scalars_bytes(n, 24)
    assert eng.msm(sc, 0) == oc.g1_mul_gen(o.poly_eval(ints(sc), tx).to_bytes(32, "big"))
    srs = eng.srs_read(0, n)
    alpha = (0xA1FA << 100) + 7
    row = rand_scalars_bytes(n, 25)
    c, ev, pf = eng.commit_open(0, row, alpha.to_bytes(32, "big"), True)
    ec = oc.commit(srs, row, True, threads=8)
    ee, ep = oc.open_(srs, row, alpha.to_bytes(32, "big"), True, threads=8)
    assert (c, ev, pf) == (ec, ee, ep)


@pytest.mark.parametrize("dist", ["all_equal", "small_32bit", "all_r_minus_1", "one_hot", "two_values"])
def test_msm_adversarial_scalar_distributions(hip, dist):
    """Structured scalars pile every digit on a few buckets: the chunked accumulate + log-depth fold must stay
    exact (and finite) for them."""
    lg, n = 14, 1 << 14
    eng = hip()
    tx = 0xBADC0DE
    eng.gen_srs(tx, 1, lg, 0)
    rnd = random.Random(3)
    if dist == "all_equal":
        sc = [0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % o.R] * n
    elif dist == "small_32bit":
        sc = [rnd.randrange(1 << 32) for _ in range(n)]
    elif dist == "all_r_minus_1":
        sc = [o.R - 1] * n
    elif dist == "one_hot":
        sc = [0] * n
        sc[n // 3] = rnd.randrange(o.R)
    else:
        sc = [(1 << 200) + 5 if i % 2 else (1 << 13) for i in range(n)]
    got = eng.msm(o.fr_to_be32(sc), 0)
    assert got == oc.g1_mul_gen(o.poly_eval(sc, tx).to_bytes(32, "big"))


@pytest.mark.parametrize("lg,window", [(8, 0), (10, 6), (12, 14), (16, 0)])
def test_msm_long_carry_runs_small_and_mid_sizes(hip, lg, window):
    """All-equal and two-valued scalars at sizes where the carry fold takes its cooperative (<= 32768 chunks) and
    its plain path, with several tree steps per bucket."""
    n = 1 << lg
    eng = hip(window)
    tx = 0xABCDE + lg
    eng.gen_srs(tx, 1, lg, 0)
    for sc in ([0x0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F % o.R] * n,
               [7 if i % 3 else o.R - 7 for i in range(n)]):
        assert eng.msm(o.fr_to_be32(sc), 0) == oc.g1_mul_gen(o.poly_eval(sc, tx).to_bytes(32, "big"))


@pytest.mark.parametrize("lg,window,spread_bits", [(17, 0, 6), (17, 18, 9), (18, 18, 9)])
def test_msm_clustered_digits_oversized_sort_partitions(hip, lg, window, spread_bits):
    """Scalars base + delta, delta < 2^spread_bits: the low window's digits fill a few ADJACENT buckets, so one or a few
    level-2 sort partitions receive far more entries than fit LDS and none of their buckets dominates -- the tile-staged
    path of k_sort_buckets, with runs above (6 bits) and below (9 bits) the whole-workgroup copy threshold; every
    other window piles on ONE bucket (the direct-scatter path).  The count-free sort overflows first and is rerun exactly."""
    n = 1 << lg
    eng = hip(window)
    tx = 0x5EED5 + lg
    eng.gen_srs(tx, 1, lg, 0)
    rnd = random.Random(lg * 100 + spread_bits)
    base = rnd.randrange(o.R >> 1) & ~((1 << 40) - 1)
    sc = [base + rnd.randrange(1 << spread_bits) for _ in range(n)]
    assert eng.msm(o.fr_to_be32(sc), 0) == oc.g1_mul_gen(o.poly_eval(sc, tx).to_bytes(32, "big"))


def test_msm_2_20_full_size_trapdoor_and_linearity(hip):
    """BASELINE.json configs[1]: 2^20-point MSM, random scalars, cached SRS.  Bit-exact against [f(tau)]G, which the
    oracle computes without any MSM; plus MSM(s) + MSM(t) == MSM(s + t) through the partial-sum ABI."""
    lg, n = 20, 1 << 20
    eng = hip()
    tx = 0x5EED5EED5EED
    eng.gen_srs(tx, 1, lg, 0)
    s_b, t_b = rand_scalars_bytes(n, 1), rand_scalars_bytes(n, 2)
    s, t = ints(s_b), ints(t_b)
    eng.upload_fr(0, s_b, False)
    got = eng.msm_resident(0, n, 0)
    assert got == oc.g1_mul_gen(o.poly_eval(s, tx).to_bytes(32, "big"))
    assert got == eng.msm(s_b, 0)
    # spot-check resident points against the oracle's independent fixed-base multiplication
    for j in (0, 1, 12345, n - 1):
        assert eng.srs_read(j, 1) == o.g1_to_be96(o.g1_table().mul(pow(tx, j, o.R)))
    u_b = o.fr_to_be32([(a + b) % o.R for a, b in zip(s, t)])
    parts = eng.msm_partial(s_b, 0) + eng.msm_partial(t_b, 0)
    assert eng.g1_sum(parts) == eng.msm(u_b, 0)
    # SRS-segment sharding as bench.py --gpus N does it: 4 shards, 4 partials, one sum
    quarter = n // 4
    shards = b"".join(eng.msm_partial(s_b[32 * k * quarter:32 * (k + 1) * quarter], k * quarter) for k in range(4))
    assert eng.g1_sum(shards) == got


def test_msm_tickets_pipeline_matches_blocking_calls(hip):
    """kzg_msm_submit / kzg_msm_wait: several requests in flight on the two lanes give exactly the blocking results
    (oracle-checked), in any interleaving; the documented E_BUSY rules hold."""
    from zkp_subnet_amd._native import KzgError, KZG_E_BUSY
    lg, n = 16, 1 << 16
    eng = hip()
    tx = 0x71C7E7
    eng.gen_srs(tx, 1, lg, 0)
    data = [rand_scalars_bytes(n, 40 + k) for k in range(3)]
    # slot 2: adversarial (all equal) so that the two lanes run different numbers of fold steps
    data[2] = data[2][:32] * n
    for k in range(3):
        eng.upload_fr(k, data[k], False)
    want = [oc.g1_mul_gen(oc.fr_eval(d, tx.to_bytes(32, "big"))) for d in data]
    assert [eng.msm_resident(k, n, 0) for k in range(3)] == want
    order = [0, 2, 1, 2, 0, 1, 1, 0, 2, 2]
    got, pending = [], []
    for k in order:
        pending.append(eng.msm_submit(k, n, 0))
        if len(pending) == 2:
            got.append(eng.msm_wait(pending.pop(0)))
    got += [eng.msm_wait(t) for t in pending]
    assert got == [want[k] for k in order]
    # sub-ranges + partial form through tickets; g1_sum is legal while a ticket is outstanding
    half = n // 2
    want_half = eng.msm(data[0][:32 * half], 0)
    ta = eng.msm_submit(0, half, 0, partial=True)
    tb = eng.msm_submit(0, half, 0, partial=True)          # same range twice: 2 * MSM(first half)
    # a blocking call made while tickets are outstanding runs on a free lane
    assert eng.msm_resident(1, n, 0) == want[1]
    tc, td = eng.msm_submit(1, n, 0), eng.msm_submit(2, n, 0)       # all four lanes now parked under tickets
    with pytest.raises(KzgError) as ei:
        eng.msm_submit(1, n, 0)
    assert ei.value.code == KZG_E_BUSY
    with pytest.raises(KzgError) as ei:
        eng.msm_resident(1, n, 0)                           # would wait forever on a single thread: refused instead
    assert ei.value.code == KZG_E_BUSY
    with pytest.raises(KzgError) as ei:
        eng.upload_fr(3, data[0], False)                    # whole-context operations need every lane idle
    assert ei.value.code == KZG_E_BUSY
    pa = eng.msm_wait(ta)
    assert eng.g1_sum(pa) == want_half                                                  # tb still outstanding
    with pytest.raises(KzgError) as ei:
        eng.upload_fr(3, data[0], False)
    assert ei.value.code == KZG_E_BUSY
    pb = eng.msm_wait(tb)
    assert eng.g1_sum(pb) == want_half          # (the 192-byte partial is a projective form: only its sum is canonical)
    assert eng.g1_sum(pa + pb) == eng.g1_sum(eng.msm_partial(data[0][:32 * half], 0) * 2)
    with pytest.raises(KzgError):
        eng.msm_wait(tb)                                                                # already collected
    assert (eng.msm_wait(td), eng.msm_wait(tc)) == (want[2], want[1])                  # any order
    eng.upload_fr(3, data[0], False)                                                   # idle again
    assert eng.msm_resident(3, n, 0) == want[0]
    eng.close()


def test_msm_ticket_cancel_frees_the_lane(hip):
    """kzg_msm_cancel: a ticket whose result will never be collected (the collective between _begin and _finish raised)
    must not park its lane forever -- afterwards exclusive calls (kzg_upload_fr) work again and results are unchanged."""
    from zkp_subnet_amd import KzgError

    eng = hip()
    n = 1 << 12
    eng.gen_srs(0xCA11CE1, 1, 12, 0)
    sc = rand_scalars_bytes(n, 4242)
    eng.upload_fr(0, sc, False)
    want = eng.msm_resident(0, n)
    tickets = [eng.msm_submit(0, n) for _ in range(4)]          # all four lanes parked
    with pytest.raises(KzgError):
        eng.upload_fr(1, sc, False)                             # E_BUSY: a ticket is outstanding
    for t in tickets[:3]:
        eng.msm_cancel(t)
    assert eng.msm_wait(tickets[3]) == want
    with pytest.raises(KzgError):
        eng.msm_cancel(tickets[0])                              # already released
    eng.upload_fr(1, sc, False)                                 # exclusive call goes through again
    assert eng.msm_resident(1, n) == want == oc.msm(eng.srs_read(0, n), sc)


def test_msm_2_24_large_size_trapdoor(hip):
    """2^24 points on one GPU (12 window tables = 26 GB resident; 2^26: test_cfg4_msm_2_26_in_eight_srs_segments):
    bit-exact against [f(tau)]G."""
    lg, n = 24, 1 << 24
    eng = hip()
    tx = 0x24242424242424242424
    eng.gen_srs(tx, 1, lg, 0)
    s_b = rand_scalars_bytes(n, 24)
    eng.upload_fr(0, s_b, False)
    got = eng.msm_resident(0, n, 0)
    y = oc.fr_eval(s_b, tx.to_bytes(32, "big"))
    assert got == oc.g1_mul_gen(y)
    half = n // 2                                               # two SRS segments, as two ranks would hold them
    parts = eng.msm_partial_resident(0, half, 0) + eng.msm_partial(s_b[32 * half:], half)
    assert eng.g1_sum(parts) == got
    eng.close()


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


@pytest.mark.parametrize("env", [{"KZG_POLY_LDS_MIN_LOG": "18"}, {"KZG_POLY_LDS_MIN_LOG": "18", "KZG_POLY_EVAL_LDS": "1"},
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


@pytest.mark.gpu
@pytest.mark.parametrize("rounds", ["1", "2"])
def test_sort_round_variants_agree(rounds):
    """Both forms of the sort's level-1 partition (one / two rounds of scalars per workgroup; the library picks by size)
    forced at sizes the oracle finishes quickly, incl. the skewed inputs that overflow a region and rerun in exact mode."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sort_variant_check.py")],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, KZG_SORT_ROUNDS=rounds))
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


# ------------------------------------------------------------------ KZG commit / open
def test_kzg_golden_vectors(hip, golden_kzg):
    tx, ty = int(golden_kzg["tau_x"], 16), int(golden_kzg["tau_y"], 16)
    for case in golden_kzg["cases"]:
        eng = hip()
        eng.gen_srs(tx, ty, case["scale"], case["machines_scale"], [case["i"]])
        row, alpha, ef = b"".join(H(v) for v in case["row"]), H(case["alpha"]), case["evaluation_form"]
        c = eng.commit(0, row, ef)
        ev, pf = eng.open(0, row, alpha, ef)
        assert (c.hex(), ev.hex(), pf.hex()) == (case["commitment"], case["eval"], case["proof"]), case["name"]
        assert eng.commit_open(0, row, alpha, ef) == (c, ev, pf), case["name"]
        eng.close()


# rows up to 2^18 take the batched two-set pass, longer ones the two-lane form (api.hip commit_open_dev): both sides of
# the switch are covered, and the fused call must equal the two separate calls (single-MSM path)
@pytest.mark.parametrize("scale,ms,i", [(10, 2, 3), (12, 0, 0), (16, 4, 9), (14, 0, 0), (18, 0, 0), (20, 1, 1)])
def test_kzg_commit_open_matches_c_oracle(hip, scale, ms, i):
    eng = hip()
    tx, ty = 0xFEEDFACE + scale, 0xDEADBEEF
    eng.gen_srs(tx, ty, scale, ms, [i])
    T = 1 << (scale - ms)
    row, alpha = rand_scalars_bytes(T, scale), rand_scalars_bytes(1, 99)
    srs = oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), scale, ms, i)
    assert eng.srs_read(0, T) == srs
    c, ev, pf = eng.commit_open(0, row, alpha, True)
    assert c == oc.commit(srs, row, True, threads=8)
    assert (ev, pf) == oc.open_(srs, row, alpha, True, threads=8)
    assert o.verify_trapdoor(tx, ty, ms, i, o.g1_decompress(c), o.g1_decompress(pf), int.from_bytes(alpha, "big"),
                             int.from_bytes(ev, "big"))
    assert eng.commit(0, row, True) == c and eng.open(0, row, alpha, True) == (ev, pf)


def test_kzg_2_22_commit_open_bit_exact(hip):
    """BASELINE.json configs[2]: degree-2^22 commit+open (Fr NTT + G1 MSM) on one GPU, bit-exact vs the CPU path:
    coefficients from the C oracle's INTT, group elements through the trapdoor identities."""
    lg, T = 22, 1 << 22
    eng = hip()
    tx = 0x7A0D007
    eng.gen_srs(tx, 1, lg, 0)
    row, alpha_b = rand_scalars_bytes(T, 22), rand_scalars_bytes(1, 23)
    alpha = int.from_bytes(alpha_b, "big")
    eng.upload_fr(1, row, True)
    c, ev, pf = eng.commit_open_resident(0, 1, T, alpha_b, True)
    coeffs_b = oc.fr_ntt(row, True)
    y = oc.fr_eval(coeffs_b, alpha_b)
    ft = int.from_bytes(oc.fr_eval(coeffs_b, tx.to_bytes(32, "big")), "big")
    assert ev == y
    assert c == oc.g1_mul_gen(ft.to_bytes(32, "big"))
    qt = (ft - int.from_bytes(y, "big")) * o.fr_inv(tx - alpha) % o.R
    assert pf == oc.g1_mul_gen(qt.to_bytes(32, "big"))
    assert eng.commit_open(0, row, alpha_b, True) == (c, ev, pf)        # host-buffer entry point agrees


def test_cfg4_msm_2_26_in_eight_srs_segments(hip):
    """BASELINE.json configs[3] on ONE GPU: a 2^26-point MSM (103 GB of window tables resident) as eight contiguous
    SRS segments of 2^23 points, one partial each, summed -- what eight ranks do with one all_gather between the partials
    and the sum -- equals the single 2^26 MSM equals [f(tau)]G from the oracle (no MSM on the CPU side)."""
    lg, n = 26, 1 << 26
    eng = hip()
    tx = 0x26262626262626262626262626
    eng.gen_srs(tx, 1, lg, 0)
    assert eng.window == 24
    seg = n // 8
    y = 0
    txs = tx.to_bytes(32, "big")
    partials = []
    tau_seg = pow(tx, seg, o.R)
    for g in range(8):
        s_b = rand_scalars_bytes(seg, 2600 + g)
        eng.upload_fr(0, s_b, False)
        partials.append(eng.msm_partial_resident(0, seg, g * seg))
        # f(tau) = sum_g tau^(g * seg) * f_g(tau)
        y = (y + pow(tau_seg, g, o.R) * int.from_bytes(oc.fr_eval(s_b, txs), "big")) % o.R
        if g == 7:
            eng.upload_fr(1, s_b, False)        # keep the last segment for the range check below
        del s_b
    want = oc.g1_mul_gen(y.to_bytes(32, "big"))
    assert eng.g1_sum(b"".join(partials)) == want
    # the same segment through the blocking compressed form == its own trapdoor value
    assert eng.msm_resident(1, seg, 7 * seg) == eng.g1_sum(partials[7])
    # one 2^26 MSM over all the scalars at once (2 GB of scalars in one slot)
    whole = b"".join(rand_scalars_bytes(seg, 2600 + g) for g in range(8))
    eng.upload_fr(2, whole, False)
    del whole
    assert eng.msm_resident(2, n, 0) == want
    eng.close()


def test_cfg5_eight_pianist_rows_2_22_on_one_engine(hip):
    """BASELINE.json configs[4] on ONE GPU: eight Pianist worker rows i = 0..7 of 2^22 coefficients, each a full
    commit+open on its own SRS slice U_i = [tau_x^j L_i(tau_y)]G (one row per GPU on an 8-GPU node; no exchange), each
    bit-exact against the trapdoor identities; the eight commitments then aggregate to the bivariate commitment."""
    lg, ms = 22, 3
    T = 1 << lg
    eng = hip()
    tx, ty = 0x5E6D5E6D5E6D5E6D, 0xA11CEA11CE
    eng.gen_srs(tx, ty, lg + ms, ms)                       # 2^25 points, 13 windows: 55 GB resident
    alpha_b = rand_scalars_bytes(1, 501)
    alpha = int.from_bytes(alpha_b, "big")
    commits, total = [], 0
    for i in range(1 << ms):
        row = rand_scalars_bytes(T, 510 + i)
        eng.upload_fr(0, row, True)
        c, ev, pf = eng.commit_open_resident(i, 0, T, alpha_b, True)
        coeffs_b = oc.fr_ntt(row, True)
        y = oc.fr_eval(coeffs_b, alpha_b)
        ft = int.from_bytes(oc.fr_eval(coeffs_b, tx.to_bytes(32, "big")), "big")
        li = o.lagrange_at(i, 1 << ms, ty)
        assert ev == y
        assert c == oc.g1_mul_gen((li * ft % o.R).to_bytes(32, "big")), i
        qt = (ft - int.from_bytes(y, "big")) * o.fr_inv(tx - alpha) % o.R
        assert pf == oc.g1_mul_gen((li * qt % o.R).to_bytes(32, "big")), i
        commits.append(c)
        total = (total + li * ft) % o.R
    assert eng.g1_sum_compressed(b"".join(commits)) == oc.g1_mul_gen(total.to_bytes(32, "big"))
    eng.close()


def test_two_processes_sharded_msm_on_one_gpu(hip):
    """Two REAL processes (fresh interpreters, torch.distributed gloo group), each with its own HipEngine on this GPU
    holding its own SRS segment, run zkp_subnet_amd.distributed.sharded_msm: the result of every rank equals the
    single-process MSM over the whole SRS and the trapdoor value."""
    import subprocess
    import sys

    lg = 17
    n = 1 << lg
    tx = 0xD157D157D157
    procs = []
    port = 29500 + (os.getpid() % 400)
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_hip_worker.py"), str(lg), hex(tx)],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-3000:]
        outs.append(so.strip().splitlines()[-1])
    assert outs[0] == outs[1]
    eng = hip()
    eng.gen_srs(tx, 1, lg + 1, 0)
    whole = rand_scalars_bytes(n, 7000) + rand_scalars_bytes(n, 7001)
    assert eng.msm(whole, 0).hex() == outs[0]
    assert outs[0] == oc.g1_mul_gen(oc.fr_eval(whole, tx.to_bytes(32, "big"))).hex()
    eng.close()


def test_stream_chained_collective_step_one_rank_rccl(hip):
    """kzg_msm_sharded_begin / _finish (the engine's lane, torch's stream with RCCL behind it, and the lane again chained
    by events: one host synchronisation per MSM) against the blocking pair and the plain MSM, in a fresh process holding
    a one-rank RCCL group (N > 1 ranks cannot share this box's single GPU)."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_rccl_onerank.py"), "29551"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])   # RCCL prints its banner after it
    assert set(rec) == {"6", "12", "16"}
    for lg, r in rec.items():
        assert r["chained_equal"] and r["blocking_equal"] and r["segment_equal"] and len(r["plain"]) == 96, (lg, r)


def test_library_collective_one_rank_no_torch(hip):
    """kzg_comm_init / kzg_msm_sharded (the all_gather is the LIBRARY's, enqueued on the lane's own stream)
    in a fresh process without torch: sharded == plain MSM == oracle, from four host threads, a forced timeout aborts the
    communicator with KZG_E_COMM inside the budget, and a rebuilt communicator serves again."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_onerank.py")], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert set(rec) == {"6", "12", "16", "20"}
    for lg, r in rec.items():
        n = 1 << int(lg)
        raw = np.random.default_rng(int(lg)).integers(0, 256, size=(n, 32), dtype=np.uint8)
        raw[:, 0] &= 0x3F
        want = oc.g1_mul_gen(oc.fr_eval(raw.tobytes(), (0x51AB1E + int(lg)).to_bytes(32, "big")))   # trapdoor: [f(tau)] G
        assert r["plain"] == want.hex(), lg
        assert r["without_comm"] is True and r["double_init"] is True, (lg, r)
        assert r["sharded_equal"] and r["segment_equal"] and r["threads_equal"], (lg, r)
        assert r["info"]["world"] == 1 and r["info"]["rank"] == 0 and r["info"]["rccl_version_code"] > 20000 and not r["info"]["broken"]
        assert r["collective_ms"] > 0 and r["world_after_destroy"] == 0, (lg, r)
    t = rec["12"]
    assert t["timeout"] is True and t["broken"] and t["after_abort"] is True and t["plain_after_abort"] and t["rebuilt_equal"], t
    # detected after the 100-ms budget; the call returns once its lane has drained (here: when the 400-ms stall kernel ends;
    # after a real abort the collective leaves the stream at once)
    assert 90 < t["timeout_after_ms"] < 3000, t


def _build_native_caller(tmp_path):
    import subprocess

    exe = str(tmp_path / "native_caller")
    libdir = os.path.join(ROOT, "zkp_subnet_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "native_caller.c"), "-o", exe, "-L", libdir, "-lkzg_mi355x", "-Wl,-rpath," + libdir,
           "-Wl,--allow-shlib-undefined"]      # (the sanitizer builds of the library resolve their runtime at load time)
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def test_c_abi_from_a_native_caller(hip, tmp_path):
    """INTEGRATION.md section 3 as a real program: tests/native_caller.c (C99, gcc, the public header, no Python or torch
    in the process) drives kzg_create / kzg_gen_srs / kzg_msm / kzg_upload_fr / kzg_msm_resident / the library's own
    collective on a one-rank communicator / kzg_commit_open; every line it prints equals the CPU oracle's answer."""
    import subprocess

    lg = 12
    n = 1 << lg
    tau = 0xC0FFEE1234567
    scal = rand_scalars_bytes(n, 4242)
    path = tmp_path / "scalars.bin"
    path.write_bytes(scal)
    exe = _build_native_caller(tmp_path)
    out = subprocess.run([exe, str(lg), tau.to_bytes(32, "big").hex(), str(path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout[-1000:], out.stderr[-2000:])
    got = dict(ln.split(" ", 1) for ln in out.stdout.splitlines() if " " in ln)
    srs = oc.srs_gen(tau.to_bytes(32, "big"), (1).to_bytes(32, "big"), lg, 0, 0)
    want = oc.msm(srs, scal).hex()
    assert got["msm"] == got["msm_resident"] == got["msm_sharded"] == want
    alpha = scal[32:64]
    assert got["commitment"] == oc.commit(srs, scal, True).hex()
    ev, pf = oc.open_(srs, scal, alpha, True)
    assert (got["eval"], got["proof"]) == (ev.hex(), pf.hex())
    assert got["comm"].startswith("rank 0 world 1 rccl 2") and got["comm"].endswith("broken 0")
    assert got["sharded_without_comm"] == "-1" and got["bad_worker_index"] == "-1" and "gfx950" in got["version"]
    assert got["multi"] == "devices 2 device_of_1 0" and got["multi_commitment"] == got["commitment"] and got["multi_bad_index"] == "-1"


def test_native_multi_device_handle_routes_rows_by_worker_index(hip):
    """kzg_multi_* (SURVEY 8b's kzg_create(device_count, device_ids)): G contexts behind one handle, worker index i served by
    context i mod G holding only the slices it serves (here G = 3 contexts on this box's one GPU, 8 worker rows).  Every
    commit / open / commit+open equals the oracle's answer for THAT worker's slice; the rows of a challenge fan out with a
    status per row -- a bad row (non-canonical scalar) costs only itself."""
    import ctypes

    from zkp_subnet_amd import _native
    from zkp_subnet_amd.engine import lagrange_factor

    lib = _native.load()
    scale, ms = 13, 3
    T, M, G = 1 << (scale - ms), 1 << ms, 3
    tx, ty = 0xABCDEF0123, 0x13579BDF
    devs = (ctypes.c_int * G)(0, 0, 0)
    m = ctypes.c_void_p()
    assert lib.kzg_multi_create(G, devs, ctypes.byref(m)) == 0
    try:
        assert lib.kzg_multi_count(m) == G and lib.kzg_multi_device_of(m, 5) == 0
        out = ctypes.create_string_buffer(48)
        assert lib.kzg_multi_commit(m, 0, bytes(32 * T), T, 1, out) == _native.KZG_E_ARG          # nothing resident yet
        s0 = b"".join(lagrange_factor(i, ms, ty).to_bytes(32, "big") for i in range(M))
        assert lib.kzg_multi_gen_srs(m, tx.to_bytes(32, "big"), s0, scale, ms) == 0
        alpha = rand_scalars_bytes(1, 9100)
        rows = [rand_scalars_bytes(T, 9000 + i) for i in range(M)]
        want = []
        for i in range(M):
            srs = oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), scale, ms, i)
            want.append((oc.commit(srs, rows[i], True),) + oc.open_(srs, rows[i], alpha, True))
        c, e, p = ctypes.create_string_buffer(48), ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
        for i in (0, 1, 2, 5, 7):
            assert lib.kzg_multi_commit(m, i, rows[i], T, 1, c) == 0 and c.raw == want[i][0], i
            assert lib.kzg_multi_open(m, i, rows[i], T, 1, alpha, e, p) == 0 and (e.raw, p.raw) == want[i][1:], i
            assert lib.kzg_multi_commit_open(m, i, rows[i], T, 1, alpha, c, e, p) == 0 and (c.raw, e.raw, p.raw) == want[i], i
        assert lib.kzg_multi_commit(m, M, rows[0], T, 1, c) == _native.KZG_E_ARG
        assert b"worker index" in lib.kzg_multi_last_error(m)
        # the rows of one challenge over all "devices" at once, in a shuffled order, one row poisoned
        order = [6, 1, 4, 7, 0, 3, 2, 5]
        idx = (ctypes.c_uint32 * M)(*order)
        blob = bytearray(b"".join(rows[i] for i in order))
        blob[32 * T * 2: 32 * T * 2 + 32] = b"\xff" * 32                       # row k = 2 (worker 4): a scalar >= r
        oc_, oe_, op_ = (ctypes.create_string_buffer(n * M) for n in (48, 32, 48))
        st = (ctypes.c_int * M)()
        rc = lib.kzg_multi_commit_open_rows(m, M, idx, bytes(blob), T, 1, alpha, oc_, oe_, op_, st)
        assert rc == _native.KZG_E_SCALAR and b"row 2 (worker 4)" in lib.kzg_multi_last_error(m)
        for k, i in enumerate(order):
            if k == 2:
                assert st[k] == _native.KZG_E_SCALAR
                continue
            assert st[k] == 0 and (oc_.raw[48 * k:48 * k + 48], oe_.raw[32 * k:32 * k + 32], op_.raw[48 * k:48 * k + 48]) == want[i], (k, i)
        # each context holds only the slices of its own workers: 8 rows over 3 contexts = 3 + 3 + 2 slices
        assert [lib.kzg_srs_points(lib.kzg_multi_ctx(m, g)) // T for g in range(G)] == [3, 3, 2]
        assert lib.kzg_multi_ctx(m, G) is None
    finally:
        lib.kzg_multi_destroy(m)


def test_g1_sum_of_k_partials_any_count(hip):
    """kzg_g1_sum over k = 1..40 partial sums (the lane-parallel tree for 2..32, the one-lane form beyond), including
    infinities (empty ranges) and repeated points (P + P inside the tree): equals the MSM over the union."""
    lg = 10
    n = 1 << lg
    eng = hip()
    tx = 0x5A5A5A5A11
    eng.gen_srs(tx, 1, lg, 0)
    sc = rand_scalars_bytes(n, 909)
    step = 25
    parts = [eng.msm_partial(sc[32 * j:32 * (j + step)], j) for j in range(0, n, step)]     # 41 partials
    for k in (1, 2, 3, 5, 7, 8, 13, 16, 17, 31, 32, 33, 40):
        m = min(n, k * step)
        assert eng.g1_sum(b"".join(parts[:k])) == oc.g1_mul_gen(oc.fr_eval(sc[:32 * m], tx.to_bytes(32, "big"))), k
    inf = eng.msm_partial(b"", 0)
    assert inf == bytes(192)
    assert eng.g1_sum(inf + parts[0] + inf + inf + parts[1]) == eng.g1_sum(parts[0] + parts[1])
    assert eng.g1_sum(inf * 5) == b"\xc0" + bytes(47)
    twice = eng.g1_sum(parts[0] * 2 + parts[1] * 2)                          # equal operands inside the tree
    want = o.g1_mul(o.g1_decompress(eng.g1_sum(parts[0] + parts[1])), 2)
    assert twice == o.g1_compress(want)
    eng.close()


def test_host_and_gpu_result_encoding_agree(hip):
    """The result point's affine conversion + compression runs on the host by default (finish_host.cpp); the GPU encoder
    (k_g1_compress[_pair], k_xyzz_pack) must give the same bytes on every entry point."""
    lg = 12
    T = 1 << lg
    eng = hip()
    eng.gen_srs(0xE2C0DE, 0x77, lg + 1, 1)
    row, alpha = rand_scalars_bytes(T, 61), rand_scalars_bytes(1, 62)
    eng.upload_fr(0, row, False)
    eng.upload_fr(1, row, True)
    got = {}
    for mode in (True, False):
        eng.set_host_finish(mode)
        t1 = eng.msm_submit(0, T, T)
        t2 = eng.msm_submit(0, T // 2, 0, partial=True)
        got[mode] = (eng.commit_open(1, row, alpha, True), eng.commit(0, row, True), eng.open(1, row, alpha, False),
                     eng.msm(row, 0), eng.msm_resident(0, T, T), eng.g1_sum(eng.msm_partial(row, 0)),
                     eng.commit_open_resident(1, 1, T, alpha, True), eng.msm_wait(t1), eng.g1_sum(eng.msm_wait(t2)),
                     eng.msm(bytes(32) * 8, 0), eng.g1_sum_compressed(eng.commit(0, row, True) * 3))
    assert got[True] == got[False]
    srs = oc.srs_gen((0xE2C0DE).to_bytes(32, "big"), (0x77).to_bytes(32, "big"), lg + 1, 1, 1)
    assert got[True][0][0] == oc.commit(srs, row, True)
    assert got[True][9] == b"\xc0" + bytes(47)
    eng.close()


def test_aggregate_commitments_and_api_commit_on_hip_engine(hip):
    """SURVEY 8a7 + 8f-4 on the HIP engine: `commit()` (reference api/commit.py:75-100) over CommitOnlyAxon(Miner) returns
    the oracle's commitment of the row, and Client.aggregate_commitments over ALL worker rows returns the commitment
    [f(tau_x, tau_y)]G of the bivariate polynomial (GPU decompression + sum; reference README.md:38)."""
    from zkp_subnet_amd import codec
    from zkp_subnet_amd.api import CommitOnlyAxon, commit
    from zkp_subnet_amd.client import Client, derive_taus
    from zkp_subnet_amd.miner import Miner, default_config

    scale, ms, seed = 9, 2, 77
    T, m = 1 << (scale - ms), 1 << ms
    miner = Miner(default_config(scale=scale, machines_scale=ms, seed=seed, setup_path=""))
    tx, ty = derive_taus(seed)
    rnd = random.Random(5)
    rows = [[rnd.randrange(o.R) for _ in range(T)] for _ in range(m)]
    axons = [CommitOnlyAxon(miner)] * 3
    comms, acc = [], 0
    for i in range(m):
        poly = [o.fr_to_b64(v) for v in rows[i]]
        got = commit(poly, axons, index=i, rng=random.Random(i))
        srs = oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), scale, ms, i)
        assert base64.b64decode(got) == oc.commit(srs, o.fr_to_be32(rows[i]), True)
        comms.append(got)
        acc = (acc + o.lagrange_at(i, m, ty) * o.poly_eval(o.ntt(rows[i], inverse=True), tx)) % o.R
    with miner.client.aggregate_commitments(comms) as r:
        assert r.status_code == 200
        assert base64.b64decode(r.json()["commitment"]) == oc.g1_mul_gen(acc.to_bytes(32, "big"))
    # -P + P = infinity; malformed / off-curve inputs are refused, not summed
    neg = bytearray(base64.b64decode(comms[0]))
    neg[0] ^= 0x20
    with miner.client.aggregate_commitments([comms[0], base64.b64encode(bytes(neg)).decode()]) as r:
        assert base64.b64decode(r.json()["commitment"]) == b"\xc0" + bytes(47)
    bad_x = bytes([0x9F]) + b"\xff" * 47                  # x >= p
    assert miner.client.aggregate_commitments([base64.b64encode(bad_x).decode()]).status_code == 400
    # ON the curve but OUTSIDE the prime-order subgroup (E(Fp) has a 2^126 cofactor): x = 5.  An untrusted miner could
    # send it as a "commitment"; the GPU membership test ([z^2]P == -sigma(P)) refuses it, alone or among valid points
    y5 = o.fp_sqrt((5 ** 3 + 4) % o.P)
    rogue = base64.b64encode(o.g1_compress((5, y5))).decode()
    assert o.is_on_curve((5, y5))
    r = miner.client.aggregate_commitments([rogue])
    assert r.status_code == 400 and "subgroup" in r.json()["error"]
    assert miner.client.aggregate_commitments(comms + [rogue]).status_code == 400
    inf = base64.b64encode(b"\xc0" + bytes(47)).decode()  # the identity IS a member
    with miner.client.aggregate_commitments([comms[1], inf]) as r:
        assert r.status_code == 200 and r.json()["commitment"] == comms[1]
    assert commit([o.fr_to_b64(1)] * T, [], index=0) == ""
    assert commit(["@@"], axons, index=0) == ""            # the miner's commit handler failed: request echoed, no string
    miner.stop()


def test_mainnet_configuration_scale_24_machines_scale_8(hip, tmp_path):
    """The reference's MAINNET prover start (Makefile:63-74: --scale 24 --machines_scale 8 from setup_24_8.uncompressed,
    2^24 points = 1.6 GB) through the production path: the setup file is written once, then `Client(setup_path).start(24,
    8)` maps and streams it (64 pinned tiles), builds the 16 window tables (34 GB); commit+open of full-length rows for
    several worker indices, each bit-exact against the trapdoor identities with L_i(tau_y) of ITS slice, verified with
    the .vk.  The start-up time is printed (and kept in profiles/ by scripts/start_time.py)."""
    import time

    from zkp_subnet_amd.client import Client, derive_taus

    scale, ms, seed = 24, 8, 2424
    T = 1 << (scale - ms)
    path = _write_setup(tmp_path, "setup_24_8.uncompressed", scale, ms, seed)
    tx, ty = derive_taus(seed)
    cl = Client(setup_path=path)
    t0 = time.perf_counter()
    cl.start(scale, ms)
    start_s = time.perf_counter() - t0
    eng = cl.engine
    print("mainnet start from file: %.2f s" % start_s, eng.load_stats())
    assert eng.srs_points == 1 << 24 and eng.window == 16
    alpha_b = rand_scalars_bytes(1, 2408)
    alpha = int.from_bytes(alpha_b, "big")
    txb = tx.to_bytes(32, "big")
    for i in (0, 1, 137, 255):
        row = rand_scalars_bytes(T, 2400 + i)
        c, ev, pf = eng.commit_open(i, row, alpha_b, True)
        coeffs_b = oc.fr_ntt(row, True)
        y = oc.fr_eval(coeffs_b, alpha_b)
        ft = int.from_bytes(oc.fr_eval(coeffs_b, txb), "big")
        li = o.lagrange_at(i, 1 << ms, ty)
        assert ev == y
        assert c == oc.g1_mul_gen((li * ft % o.R).to_bytes(32, "big")), i
        qt = (ft - int.from_bytes(y, "big")) * o.fr_inv(tx - alpha) % o.R
        assert pf == oc.g1_mul_gen((li * qt % o.R).to_bytes(32, "big")), i
        assert eng.verify(i, pf, alpha_b, ev, c)
        assert not eng.verify((i + 7) % 256, pf, alpha_b, ev, c)
    # spot-check resident points of the last slice against the oracle's fixed-base multiplication
    for j in (0, T - 1):
        want = o.g1_table().mul(pow(tx, j, o.R) * o.lagrange_at(255, 1 << ms, ty) % o.R)
        assert eng.srs_read(255 * T + j, 1) == o.g1_to_be96(want)
    cl.stop()
    os.remove(path)
    os.remove(path + ".vk")


def test_client_and_miner_on_hip_engine(hip, fr_kat):
    """The reference miner test (tests/test_miner.py:62-121) on the HIP engine: 16-coefficient TEST_POLY at
    scale 6 / machines_scale 2; forward() returns the client's commitment and proof; oracle agrees bit for bit."""
    from zkp_subnet_amd import codec
    from zkp_subnet_amd.client import Client, derive_taus
    from zkp_subnet_amd.miner import Miner, default_config
    from zkp_subnet_amd.protocol import Prove

    client = Client(port=1337, bin="./test_prover", setup_path="test_setup.compressed",
                    precompute_path="test_precompute.compressed", seed=6)
    miner = Miner(default_config(scale=6, machines_scale=2, seed=6), client=client)
    syn = Prove(index=0, poly=fr_kat["poly"], alpha=fr_kat["point"], eval=fr_kat["eval"])
    with miner.client.worker_commit(i=0, poly=syn.poly) as r:
        assert r.status_code == 200
        commitment = r.json()["commitment"]
    with miner.client.worker_open(i=0, poly=syn.poly, x=syn.alpha) as r:
        assert r.status_code == 200
        ev, proof = r.json()["eval"], r.json()["proof"]
    with miner.client.worker_verify(i=0, proof=proof, alpha=syn.alpha, eval=ev, commitment=commitment) as r:
        assert r.status_code == 200 and r.json().get("valid") is True       # reference tests/test_miner.py:101-111
    raw = base64.b64decode(proof)                                            # reference tests/test_validator.py:79-86
    bumped = base64.b64encode((int.from_bytes(raw, "big") + 1).to_bytes(len(raw), "big")).decode()
    with miner.client.worker_verify(i=0, proof=bumped, alpha=syn.alpha, eval=ev, commitment=commitment) as r:
        assert r.status_code == 200 and r.json().get("valid") is False
    with miner.client.worker_verify(i=1, proof=proof, alpha=syn.alpha, eval=ev, commitment=commitment) as r:
        assert r.json().get("valid") is False                               # another worker's basis
    ret = miner.forward(syn)
    assert (ret.commitment, ret.proof, ret.eval) == (commitment, proof, ev)
    tx, ty = derive_taus(6)
    srs = oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), 6, 2, 0)
    row = codec.fr_list_to_be32(syn.poly)
    assert codec.g1_from_b64(commitment) == oc.commit(srs, row, True)
    assert (codec.fr_to_be32(ev), codec.g1_from_b64(proof)) == oc.open_(srs, row, codec.fr_to_be32(syn.alpha), True)
    assert miner.client.worker_commit(i=0, poly=["bad"]).status_code == 400
    too_big = base64.b64encode((o.R + 5).to_bytes(32, "big")).decode().rstrip("=")
    assert miner.client.worker_commit(i=0, poly=[too_big] * 16).status_code == 400                 # Fr >= r
    with miner.client.fft(syn.poly, left=True, inverse=True) as r:
        coeffs = r.json()["poly"]
    with miner.client.eval(coeffs, syn.alpha) as r:
        assert r.json()["y"] == ev
    # validator mirror end to end on the HIP engine: challenge -> forward -> pairing-verified reward table
    from zkp_subnet_amd.validator import generate_challenge, reward

    ch = generate_challenge(miner.client, 2)
    responses = [miner.forward(ch.to_synapse(i)) for i in range(2)]
    assert [reward(miner.client, ch, responses[i], i, 0.0) for i in range(2)] == [1.0, 1.0]
    assert reward(miner.client, ch, responses[0], 0, 15.0) == 0.5
    assert reward(miner.client, ch, responses[0], 1, 0.0) == 0.0
    assert responses[1].eval == ch.evals[1]
    # the fused validator step (one call, coefficients stay on the device) == the reference's two calls; all rows of a
    # step verified on a thread pool; random_poly / random_point come from the native generator, uniform below r
    from zkp_subnet_amd.validator import verify_all

    with miner.client.fft_eval(syn.poly, syn.alpha, left=True, inverse=True) as r:
        assert r.status_code == 200 and r.json()["y"] == ev
    ch4 = generate_challenge(miner.client, 4)
    for i in range(4):
        with miner.client.fft(ch4.polys[i], left=True, inverse=True) as r:
            cf = r.json()["poly"]
        with miner.client.eval(cf, ch4.alpha) as r:
            assert r.json()["y"] == ch4.evals[i]
    resp4 = [miner.forward(ch4.to_synapse(i)) for i in range(4)]
    assert verify_all(miner.client, ch4, resp4, threads=4) == [True] * 4
    resp4[2] = resp4[2].model_copy(update={"proof": resp4[1].proof})
    assert verify_all(miner.client, ch4, resp4 + [None], threads=4)[:4] == [True, True, False, True]
    with miner.client.random_poly() as r:
        rp = r.json()["poly"]
    assert len(rp) == 4 and all(len(row) == 16 for row in rp)
    assert all(int.from_bytes(codec.fr_to_be32(s), "big") < o.R for row in rp for s in row)
    miner.stop()


def test_concurrent_host_threads_and_contexts(hip):
    """The axon calls forward() from worker threads (SURVEY 8b threading): four threads hammer ONE context (each call on
    its own lane and pinned staging buffer, running concurrently on the GPU) while a fifth drives a second context on the
    same GPU; every answer equals the oracle's."""
    import threading

    from zkp_subnet_amd import codec
    from zkp_subnet_amd.client import Client, derive_taus

    lg = 10
    T = 1 << lg
    cl = Client(seed=21, workers=[0, 1])
    cl.start(scale=lg + 1, machines_scale=1)
    other = Client(seed=22, workers=[0])
    other.start(scale=lg, machines_scale=0)
    rows = [rand_scalars_bytes(T, 300 + k) for k in range(6)]
    alphas = [rand_scalars_bytes(1, 400 + k) for k in range(6)]
    want = {}
    for c_, seed, ms in ((cl, 21, 1), (other, 22, 0)):
        tx, ty = (t.to_bytes(32, "big") for t in derive_taus(seed))
        for w in ((0, 1) if c_ is cl else (0,)):
            srs = oc.srs_gen(tx, ty, lg + ms, ms, w)
            for k in range(6):
                ev, pf = oc.open_(srs, rows[k], alphas[k], True)
                want[(id(c_), w, k)] = (codec.g1_to_b64(oc.commit(srs, rows[k], True)), codec.be32_to_fr(ev), codec.g1_to_b64(pf))
    polys = [codec.be32_to_fr_list(r) for r in rows]
    xs = [codec.be32_to_fr(a) for a in alphas]
    errors = []

    def work(c_, workers, tid):
        try:
            for it in range(6):
                k, w = (it + tid) % 6, workers[(it + tid) % len(workers)]
                with c_.worker_commit_and_open(w, polys[k], xs[k]) as r:
                    b = r.json()
                    if r.status_code != 200 or (b["commitment"], b["eval"], b["proof"]) != want[(id(c_), w, k)]:
                        errors.append((tid, it, r.status_code))
                with c_.worker_commit(w, polys[k]) as r:
                    if r.json().get("commitment") != want[(id(c_), w, k)][0]:
                        errors.append((tid, it, "commit"))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(cl, [0, 1], t)) for t in range(4)]
    threads.append(threading.Thread(target=work, args=(other, [0], 4)))
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    cl.stop()
    other.stop()


def test_compressed_srs_load_and_read(hip):
    """ZCash-compressed setup files (reference `uncompressed=False`, base/miner.py:75-81): the GPU recovers every y by
    a square root; result identical to loading the uncompressed points; malformed encodings fail the load."""
    from zkp_subnet_amd._native import KzgError, KZG_E_POINT
    tx = 0xC0FFEE
    n = 256
    pts = [o.g1_table().mul(pow(tx, j, o.R)) for j in range(n - 1)] + [None]         # last record: infinity
    unc = b"".join(o.g1_to_be96(p) for p in pts)
    cmp_ = b"".join(o.g1_compress(p) for p in pts)
    assert sum(c[0] & 0x20 != 0 for c in (cmp_[48 * k:48 * k + 48] for k in range(n))) > 20   # both y signs occur
    a, b = hip(), hip()
    a.load_srs(unc, 8, 0)
    b.load_srs(cmp_, 8, 0, compressed=True)
    assert b.srs_read(0, n) == unc == a.srs_read(0, n)
    assert a.srs_read(0, n, compressed=True) == cmp_ == b.srs_read(0, n, compressed=True)
    s_b = rand_scalars_bytes(n, 91)
    assert a.msm(s_b, 0) == b.msm(s_b, 0) == oc.msm(unc[:96 * (n - 1)], s_b[:32 * (n - 1)])   # infinity adds nothing
    # an x with no point above it, an unreduced x, a missing compression flag, a dirty infinity
    x = 1
    while o.fp_sqrt((x ** 3 + 4) % o.P) is not None:
        x += 1
    no_point = bytearray(x.to_bytes(48, "big")); no_point[0] |= 0x80
    unreduced = bytearray(o.P.to_bytes(48, "big")); unreduced[0] |= 0x80
    no_flag = bytearray(cmp_[:48]); no_flag[0] &= 0x7F
    dirty_inf = bytearray(48); dirty_inf[0] = 0xC0; dirty_inf[47] = 1
    for bad in (no_point, unreduced, no_flag, dirty_inf):
        with pytest.raises(KzgError) as ei:
            hip().load_srs(bytes(bad) + cmp_[48:], 8, 0, compressed=True)
        assert ei.value.code == KZG_E_POINT
    a.close(); b.close()


def test_setup_cli_file_roundtrip_through_client(hip, tmp_path, fr_kat):
    """`setup` writes the SRS + verifier key files on the GPU; a fresh Client loads them (the reference's
    tests/conftest.py:50-65 flow) and commit / open / verify agree with a Client that generated the same SRS in memory."""
    from zkp_subnet_amd import setup_cli
    from zkp_subnet_amd.client import Client

    path = str(tmp_path / "test_setup.uncompressed")
    assert setup_cli.main(["setup", "--setup-path", path, "--precompute-path", path + ".pre", "--scale", "6",
                           "--machines-scale", "2", "--generate-setup", "--generate-precompute", "--overwrite",
                           "--seed", "42"]) == 0
    assert os.path.getsize(path) == 64 * 96 and os.path.getsize(path + ".vk") == 192 + 4 * 96
    from_file = Client(setup_path=path, precompute_path=path + ".pre")
    from_file.start(scale=6, machines_scale=2)
    in_memory = Client(seed=42)
    in_memory.start(scale=6, machines_scale=2)
    for i in (0, 3):
        with from_file.worker_commit_and_open(i, fr_kat["poly"], fr_kat["point"]) as a, \
                in_memory.worker_commit_and_open(i, fr_kat["poly"], fr_kat["point"]) as b:
            assert a.status_code == 200 and a.json() == b.json()
            body = a.json()
        with from_file.worker_verify(i, body["proof"], fr_kat["point"], body["eval"], body["commitment"]) as r:
            assert r.json()["valid"] is True
        with from_file.worker_verify((i + 1) % 4, body["proof"], fr_kat["point"], body["eval"], body["commitment"]) as r:
            assert r.json()["valid"] is False
    # the same setup written compressed and loaded with the reference's uncompressed=False flag
    cpath = str(tmp_path / "test_setup.compressed")
    assert setup_cli.main(["setup", "--setup-path", cpath, "--scale", "6", "--machines-scale", "2", "--generate-setup",
                           "--compressed", "--seed", "42"]) == 0
    assert os.path.getsize(cpath) == 64 * 48
    from_c = Client(setup_path=cpath, uncompressed=False)
    from_c.start(scale=6, machines_scale=2)
    with from_c.worker_commit_and_open(3, fr_kat["poly"], fr_kat["point"]) as c:
        assert c.status_code == 200 and c.json() == body
    with from_c.worker_verify(3, body["proof"], fr_kat["point"], body["eval"], body["commitment"]) as r:
        assert r.json()["valid"] is True
    from_c.stop()
    from_file.stop()
    in_memory.stop()


def test_two_call_route_is_served_from_the_row_cache_only_for_the_same_content(hip):
    """The UNCHANGED reference miner calls worker_commit(i, poly) then worker_open(i, poly, x) (neurons/miner.py:56-61).
    The second call finds the row's coefficient vector on the device (keyed by the 128-bit content tag of the decoded
    bytes): no upload, no INTT -- and the results are the oracle's.  A row with ONE changed coefficient between the two
    calls must NOT be served from the cache."""
    import threading

    from zkp_subnet_amd.client import Client, derive_taus

    scale, ms, seed = 12, 2, 4321
    T = 1 << (scale - ms)
    cl = Client(seed=seed)
    cl.start(scale, ms)
    eng = cl.engine
    tx, ty = derive_taus(seed)
    srs = {i: oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), scale, ms, i) for i in range(4)}
    rnd = random.Random(99)
    alpha = rnd.randrange(o.R)
    alpha_s, alpha_b = o.fr_to_b64(alpha), alpha.to_bytes(32, "big")

    def expect(i, vals):
        rb = o.fr_to_be32(vals)
        ev, pf = oc.open_(srs[i], rb, alpha_b, True)
        return oc.commit(srs[i], rb, True), ev, pf

    def commit(i, poly):
        with cl.worker_commit(i, poly) as r:
            assert r.status_code == 200
            return base64.b64decode(r.json()["commitment"])

    def open_(i, poly):
        with cl.worker_open(i, poly, alpha_s) as r:
            assert r.status_code == 200
            return o.fr_from_b64(r.json()["eval"]).to_bytes(32, "big"), base64.b64decode(r.json()["proof"])

    vals = [rnd.randrange(o.R) for _ in range(T)]
    poly = [o.fr_to_b64(v) for v in vals]
    h0, m0 = eng.row_cache_stats()
    want = expect(1, vals)
    assert commit(1, poly) == want[0]
    assert eng.row_cache_stats() == (h0, m0 + 1)                       # first sight of the row: a miss, now cached
    assert open_(1, poly) == want[1:]
    assert eng.row_cache_stats() == (h0 + 1, m0 + 1)                   # the open was served from the cache
    assert open_(2, list(poly)) == expect(2, vals)[1:]                 # same content, other list object, other worker
    assert eng.row_cache_stats() == (h0 + 2, m0 + 1)
    # ONE coefficient changed between commit and open: not the cached row
    vals2 = list(vals)
    vals2[T // 2] = (vals2[T // 2] + 1) % o.R
    poly2 = list(poly)
    poly2[T // 2] = o.fr_to_b64(vals2[T // 2])
    assert open_(1, poly2) == expect(1, vals2)[1:] != want[1:]
    assert eng.row_cache_stats() == (h0 + 2, m0 + 2)
    assert open_(1, poly) == want[1:]                                  # the original row is still cached
    assert eng.row_cache_stats() == (h0 + 3, m0 + 2)
    # a failed call leaves nothing behind: a non-canonical scalar (>= r) is refused again on the retry
    bad = list(poly)
    bad[3] = base64.b64encode(o.R.to_bytes(32, "big")).decode().rstrip("=")
    assert cl.worker_commit(1, bad).status_code == 400
    assert cl.worker_open(1, bad, alpha_s).status_code == 400
    # more distinct rows than cache slots, then the first again: evicted, recomputed, still right
    for k in range(6):
        vk = [rnd.randrange(o.R) for _ in range(T)]
        pk = [o.fr_to_b64(v) for v in vk]
        wk = expect(k % 4, vk)
        assert commit(k % 4, pk) == wk[0] and open_(k % 4, pk) == wk[1:]
    assert commit(1, poly) == want[0] and open_(1, poly) == want[1:]
    # shorter prefix of the same row: other length, other tag
    assert open_(3, poly[: T // 2]) == expect(3, vals[: T // 2])[1:]
    # the axon's worker threads: the same and different rows concurrently
    errors = []

    def worker(t):
        try:
            r2 = random.Random(500 + t)
            for it in range(6):
                if it % 2:
                    assert commit(1, poly) == want[0] and open_(1, poly) == want[1:]
                else:
                    v = [r2.randrange(o.R) for _ in range(T)]
                    pv = [o.fr_to_b64(x) for x in v]
                    w = expect(t % 4, v)
                    assert commit(t % 4, pv) == w[0] and open_(t % 4, pv) == w[1:]
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    cl.stop()


def _write_setup(tmp_path, name, scale, ms, seed, compressed=False):
    import shutil

    from zkp_subnet_amd import setup_cli

    need = ((48 if compressed else 96) << scale) + (1 << 20)
    free = shutil.disk_usage(str(tmp_path)).free
    if free < 2 * need:          # a full scratch disk must not take the whole suite down (-x): say so and skip this one
        pytest.skip(f"setup file of {need >> 20} MiB needs scratch space, only {free >> 20} MiB free under {tmp_path}")
    path = str(tmp_path / name)
    args = ["setup", "--setup-path", path, "--scale", str(scale), "--machines-scale", str(ms), "--generate-setup",
            "--overwrite", "--seed", str(seed)] + (["--compressed"] if compressed else [])
    assert setup_cli.main(args) == 0
    assert os.path.getsize(path) == (48 if compressed else 96) << scale
    assert os.path.getsize(path + ".vk") == 192 + 96 * (1 << ms)
    return path


def test_production_start_testnet_20_8_from_setup_files(hip, tmp_path):
    """The reference's TESTNET start path (Makefile:89-101: --scale 20 --machines_scale 8) as the reference runs it:
    `Client(setup_path=...).start(20, 8)` from a setup FILE (base/miner.py:75-84) -- not gen_srs -- uncompressed and with
    uncompressed=False, with its .vk; commit / open / fused commit+open for workers 0, 137, 255 == the C oracle on that
    worker's slice; worker_verify true, false for another worker."""
    from zkp_subnet_amd.client import Client, derive_taus

    scale, ms, seed = 20, 8, 2008
    T = 1 << (scale - ms)
    tx, ty = derive_taus(seed)
    txb, tyb = tx.to_bytes(32, "big"), ty.to_bytes(32, "big")
    paths = [(_write_setup(tmp_path, "setup_20_8.uncompressed", scale, ms, seed), True),
             (_write_setup(tmp_path, "setup_20_8.compressed", scale, ms, seed, compressed=True), False)]
    rnd = random.Random(208)
    alpha = rnd.randrange(o.R)
    alpha_s, alpha_b = o.fr_to_b64(alpha), alpha.to_bytes(32, "big")
    seen = {}
    for path, unc in paths:
        cl = Client(setup_path=path, uncompressed=unc)
        cl.start(scale, ms)
        assert cl.engine.srs_points == 1 << scale and cl.engine.window == 12
        st = cl.engine.load_stats()
        assert st["total_s"] > 0 and st["tables_s"] > 0
        for i in (0, 137, 255):
            row = [random.Random(1000 + i).randrange(o.R) for _ in range(T)]
            poly, row_b = [o.fr_to_b64(v) for v in row], o.fr_to_be32(row)
            srs = oc.srs_gen(txb, tyb, scale, ms, i)
            assert cl.engine.srs_read(i * T, T) == srs
            want_c = oc.commit(srs, row_b, True)
            want_e, want_p = oc.open_(srs, row_b, alpha_b, True)
            with cl.worker_commit(i, poly) as r:
                assert r.status_code == 200 and base64.b64decode(r.json()["commitment"]) == want_c
            with cl.worker_open(i, poly, alpha_s) as r:
                assert o.fr_from_b64(r.json()["eval"]) == int.from_bytes(want_e, "big")
                assert base64.b64decode(r.json()["proof"]) == want_p
            with cl.worker_commit_and_open(i, poly, alpha_s) as r:
                body = r.json()
                assert base64.b64decode(body["commitment"]) == want_c and base64.b64decode(body["proof"]) == want_p
            with cl.worker_verify(i, body["proof"], alpha_s, body["eval"], body["commitment"]) as r:
                assert r.json()["valid"] is True
            with cl.worker_verify((i + 1) % 256, body["proof"], alpha_s, body["eval"], body["commitment"]) as r:
                assert r.json()["valid"] is False
            assert seen.setdefault(i, body) == body          # compressed and uncompressed files: identical answers
        cl.stop()
    for path, _ in paths:
        os.remove(path)
        os.remove(path + ".vk")


def test_multi_tile_setup_file_2_22_boundaries_and_rollback(hip, tmp_path):
    """A 2^22-point setup file is streamed through 16 pinned tiles of 2^18 points: the resident table equals a gen_srs
    engine's at every tile boundary (window 0 and the highest window), for the uncompressed AND the compressed file; ONE
    bad point in tile 3 fails the reload with KZG_E_POINT and the previously loaded SRS keeps serving (tables are built
    aside and swapped in only on success)."""
    from zkp_subnet_amd._native import KzgError, KZG_E_POINT
    from zkp_subnet_amd.client import derive_taus

    scale, ms, seed = 22, 8, 2208
    T, tile = 1 << (scale - ms), 1 << 18
    tx, ty = derive_taus(seed)
    path = _write_setup(tmp_path, "setup_22_8.uncompressed", scale, ms, seed)
    cpath = _write_setup(tmp_path, "setup_22_8.compressed", scale, ms, seed, compressed=True)
    ref = hip()
    ref.gen_srs(tx, ty, scale, ms)
    wtop = len(ref.window_offsets) - 2
    probes = sorted({max(0, k * tile + d) for k in range(17) for d in (-2, -1, 0, 1)} & set(range(1 << scale)))
    want = {(w, j): ref.srs_read(j, 1, window=w) for w in (0, wtop) for j in probes}
    eng = hip()
    for pth, comp in ((path, False), (cpath, True)):
        eng.load_srs_file(pth, scale, ms, compressed=comp)
        assert eng.srs_points == 1 << scale
        for (w, j), v in want.items():
            assert eng.srs_read(j, 1, window=w) == v, (comp, w, j)
    row = rand_scalars_bytes(T, 2209)
    alpha_b = rand_scalars_bytes(1, 2210)
    before = eng.commit_open(200, row, alpha_b, True)
    assert before == ref.commit_open(200, row, alpha_b, True)
    # one bad point in tile 3: y off the curve (uncompressed) / an x with no point above it (compressed)
    bad_at = 3 * tile + 5
    with open(path, "r+b") as f:
        f.seek(96 * bad_at + 95)
        last = f.read(1)
        f.seek(96 * bad_at + 95)
        f.write(bytes([last[0] ^ 1]))
    x = 1
    while o.fp_sqrt((x ** 3 + 4) % o.P) is not None:
        x += 1
    no_point = bytearray(x.to_bytes(48, "big"))
    no_point[0] |= 0x80
    with open(cpath, "r+b") as f:
        f.seek(48 * bad_at)
        f.write(bytes(no_point))
    for pth, comp in ((path, False), (cpath, True)):
        with pytest.raises(KzgError) as ei:
            eng.load_srs_file(pth, scale, ms, compressed=comp)
        assert ei.value.code == KZG_E_POINT
        assert eng.srs_points == 1 << scale                      # rollback: the previous table still serves
        assert eng.commit_open(200, row, alpha_b, True) == before
    with pytest.raises(KzgError):
        eng.load_srs_file(str(tmp_path / "absent"), scale, ms)
    assert eng.commit_open(200, row, alpha_b, True) == before
    ref.close()
    eng.close()
    for pth in (path, cpath):
        os.remove(pth)
        os.remove(pth + ".vk")


def test_bench_contract_line(hip):
    """bench.py prints ONE JSON line, last on stdout, with the driver's keys plus `roofline` and `cpu_baseline`."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--log-n", "14", "--steps", "3", "--warmup", "1",
                          "--cpu-sample-log", "12", "--kzg-rows", "10,8", "--e2e-rows", "10,8"], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert sum(1 for ln in lines if ln.startswith("{")) == 1            # N = 1: exactly ONE JSON line
    rec = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["higher_is_better"] is True
    assert rec["scaling"] == "weak" and rec["data"] == "synthetic" and "workload" in rec["config"]
    assert abs(rec["value"] - (1 << 14) * 3 / (rec["ms_per_step"] * 3e-3)) / rec["value"] < 1e-6
    rf, cb = rec["roofline"], rec["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["matches_gpu_bit_exact"] is True
    assert "1" in cb["points_per_s_by_threads"] and len(cb["points_per_s_by_threads"]) >= 2      # 1 thread AND more
    assert rec["pipelined"]["value"] > 0 and rec["pipelined"]["pre_warm_steps"] >= 1
    assert rec["pre_warm_steps"] >= rec["pipelined"]["pre_warm_steps"] + 3      # what ran before the W declared warm-up steps
    # the line says who measured it (VERDICT r4 task 4): library, bench.py and the whole source set
    import hashlib
    from bench import source_sha16
    ident = rec["identity"]
    assert "gfx950" in ident["lib_version"] and ident["source_sha16"] == source_sha16(root)
    assert ident["bench_py_sha16"] == hashlib.sha256(open(os.path.join(root, "bench.py"), "rb").read()).hexdigest()[:16]
    # the reference's own route from text at the reference's sizes (task 5): two calls + the fused call, each == C oracle
    for key in ("2^10", "2^8"):
        e2e = rec["e2e_from_text"][key]
        assert e2e["matches_cpu_oracle_bit_exact"] is True and e2e["two_call_ms"]["requests"] >= 20
        assert 0 < e2e["fused_ms"]["median"] <= e2e["two_call_ms"]["p90"] * 1.5
        assert e2e["two_call_row_cache_hits_misses"] == [30, 30]        # every worker_open a verified hit, every commit a miss
    for key in ("2^10", "2^8"):                     # commit+open latency rows, each with roofline + cpu_baseline
        row = rec["kzg_commit_open"][key]
        assert row["ms"] > 0 and row["p10"] <= row["ms"] <= row["p90"] and row["roofline"]["algorithmic_bytes"] == 384.0 * (1 << row["log2_T"])
        assert row["cpu_baseline"]["matches_gpu_bit_exact"] is True and len(row["result_hex"]) == 2 * (48 + 32 + 48)
    # the collective path (1-rank group; the all_gather is the LIBRARY's: kzg_comm_init / kzg_msm_sharded on real RCCL)
    # gives the same point
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--log-n", "14", "--steps", "3", "--warmup", "1",
                           "--no-cpu-baseline", "--no-adversarial", "--no-kzg-rows", "--msm26-log", "16", "--kzg22-log", "12"],
                          capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out2.returncode == 0, out2.stderr[-2000:]
    jl = [json.loads(ln) for ln in out2.stdout.splitlines() if ln.strip().startswith("{")]
    # with a process group the headline is printed as soon as it is complete, the augmented line follows: last line wins
    assert len(jl) == 2 and "partial_line" in jl[0] and "msm26" not in jl[0] and "partial_line" not in jl[1]
    assert jl[0]["value"] == jl[1]["value"] and jl[0]["result_hex"] == jl[1]["result_hex"]
    rec2 = jl[-1]
    assert rec2["result_hex"] == rec["result_hex"] and len(rec["result_hex"]) == 96
    # a process group + no --workload: the same launch also yields configs[3] (msm26) and configs[4] (pianist_kzg22)
    assert rec2["config"]["world_size"] == 1 and rec2["config"]["rccl_version"].startswith("2.")
    assert rec2["config"]["collective"].startswith("library: ncclAllGather") and "comm_init_s" in rec2["config"]["collective_detail"]
    m26, pk = rec2["msm26"], rec2["pianist_kzg22"]
    assert m26["scaling"] == "strong" and m26["all_ranks_equal"] and m26["value"] > 0 and m26["roofline"]["kernel_ms"] > 0
    assert abs(m26["value"] - (1 << 16) * m26["steps"] / (m26["ms_per_step"] * m26["steps"] * 1e-3)) / m26["value"] < 1e-6
    assert pk["scaling"] == "weak" and pk["value"] > 0 and len(pk["results_hex_by_rank"]) == 1
    assert pk["aggregate_commitment_hex"] == pk["results_hex_by_rank"][0][:96]      # one row: the sum is the row's own
    # ... and both agree with the oracle on the same seeded inputs
    from bench import TAU, uniform_fr
    e = hip()
    e.gen_srs(TAU, 1, 16, 0)
    assert bytes.fromhex(m26["result_hex"]) == oc.msm(e.srs_read(0, 1 << 16), uniform_fr(1 << 16, 1000), threads=4)
    e.gen_srs(TAU, 0, 12, 0, factors=[1])
    row, alpha = uniform_fr(1 << 12, 0), uniform_fr(1, 1)
    srs = e.srs_read(0, 1 << 12)
    want = oc.commit(srs, row, True) + b"".join(oc.open_(srs, row, alpha, True))
    assert bytes.fromhex(pk["results_hex_by_rank"][0]) == want
    assert rec2["pipelined"]["value"] > 0
    # a communicator that cannot be built (injected on the only rank): every rank falls back to the process group's
    # all_gather, the line says so, the result is the same
    env3 = dict(env, BENCH_FAULT="comm_init:0", MASTER_PORT="29548")
    out3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--log-n", "14", "--steps", "3", "--warmup", "1",
                           "--no-cpu-baseline", "--no-adversarial", "--no-dist-extra", "--no-pipelined"],
                          capture_output=True, text=True, timeout=600, cwd=root, env=env3)
    assert out3.returncode == 0, out3.stderr[-2000:]
    rec3 = json.loads([ln for ln in out3.stdout.splitlines() if ln.strip().startswith("{")][-1])
    assert rec3["result_hex"] == rec["result_hex"]
    assert rec3["config"]["collective"].startswith("gloo fallback (library RCCL preflight failed on rank 0: Fault")
    assert rec3["config"]["rccl_version"].startswith("none (gloo fallback")


def test_bench_pipelined_region_overlaps_two_requests(hip):
    """Two MSMs in flight must buy something over one at a time (a copy stream created between `aux` and
    the lanes' streams had put lanes 0 and 1 on one hardware queue -- 2.64-2.73 ms against 2.44-2.46, profiles/
    r05_ab_pipelined_bisect.log).  Full size, the driver's K / W; the bound is loose (<= 1.02 x serial) so that only the
    loss of the overlap trips it, not a box's noise."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                          "--no-kzg-rows", "--no-adversarial", "--no-e2e"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][-1])
    assert rec["config"]["points_per_gpu"] == 1 << 20
    assert rec["pipelined"]["ms_per_step"] <= 1.02 * rec["ms_per_step"], (rec["pipelined"], rec["ms_per_step"])


def test_bench_two_ranks_on_one_gpu_exercises_the_multi_rank_logic(hip):
    """What a driver SCALE launch runs, with TWO real ranks: `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`.
    A one-GPU box cannot form an RCCL group of two, so both ranks share device 0 and exchange their partials through gloo
    (BENCH_ONE_GPU / BENCH_BACKEND: self-test knobs): everything rank-dependent is real -- SRS segment r of the 2 n-point
    SRS per rank, scalars of its index range, partial -> all_gather -> sum, the cross-rank equality checks, and the
    `msm26` (strong scaling) and `pianist_kzg22` (one row per rank) objects of the same launch.  All three results are
    compared with the oracle on the concatenated inputs."""
    import subprocess
    import sys

    from bench import TAU, R_MOD, uniform_fr
    from zkp_subnet_amd.engine import lagrange_factor

    env = dict(os.environ, BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", BENCH_COMM_INIT_TIMEOUT_S="60")
    env.pop("BENCH_BACKEND", None)
    args = ["--gpus", "2", "--log-n", "13", "--steps", "3", "--warmup", "1", "--msm26-log", "15", "--kzg22-log", "11",
            "--cpu-sample-log", "12"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    jl = [json.loads(ln) for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(jl) == 2 and "partial_line" in jl[0] and "msm26" in jl[1]     # headline first, the augmented line last
    rec = jl[-1]
    # RCCL refuses two ranks on one device: the library's preflight fails on every rank, all of them fall back to the
    # gloo group's all_gather and the line says so -- a forced RCCL-init failure still yields a line (VERDICT r4 task 1c)
    assert rec["config"]["collective"].startswith("gloo fallback (library RCCL preflight failed on rank 0")
    assert set(rec["config"]["collective_detail"]["library_preflight_failed"]) == {"0", "1"}
    assert rec["config"]["rccl_version"].startswith("none (gloo fallback")
    # ... and the SAME line from `python bench.py --gpus 2` with NO launcher: the parent starts that launch line itself
    # as a child before anything touches the GPU, relays it, and the JSON line is the last line of stdout
    env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=900, cwd=ROOT, env=env2)
    assert out2.returncode == 0, out2.stderr[-3000:]
    assert sum(1 for ln in out2.stdout.splitlines() if ln.strip().startswith("{")) == 1    # the parent relays the LAST line only
    rec2 = json.loads([ln for ln in out2.stdout.splitlines() if ln.strip()][-1])
    assert rec2["n_gpus"] == 2 and rec2["config"]["world_size"] == 2 and "msm26" in rec2
    for k in ("result_hex", "metric", "unit", "scaling", "steps", "warmup"):
        assert rec2[k] == rec[k], k
    assert rec2["msm26"]["result_hex"] == rec["msm26"]["result_hex"]
    assert rec2["pianist_kzg22"]["results_hex_by_rank"] == rec["pianist_kzg22"]["results_hex_by_rank"]
    for r_ in (rec, rec2):                          # rule (d) on an N > 1 line: roofline AND cpu_baseline
        cb = r_["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["matches_gpu_bit_exact"] is True
        assert r_["roofline"]["achieved"] > 0
    assert rec["n_gpus"] == 2 and rec["config"]["world_size"] == 2 and rec["scaling"] == "weak"
    assert abs(rec["value"] - 2 * (1 << 13) * 3 / (rec["ms_per_step"] * 3e-3)) / rec["value"] < 1e-6
    e = hip()
    # headline: 2 x 2^13 points, rank r = segment r, scalars seeded by the rank
    e.gen_srs(TAU, 1, 14, 0)
    want = oc.msm(e.srs_read(0, 1 << 14), uniform_fr(1 << 13, 0) + uniform_fr(1 << 13, 1), threads=4)
    assert bytes.fromhex(rec["result_hex"]) == want
    # msm26 object: ONE 2^15-point MSM in two segments
    m26 = rec["msm26"]
    assert m26["n_gpus"] == 2 and m26["points_per_gpu"] == 1 << 14 and m26["all_ranks_equal"] and m26["scaling"] == "strong"
    e.gen_srs(TAU, 1, 15, 0)
    want = oc.msm(e.srs_read(0, 1 << 15), uniform_fr(1 << 14, 1000) + uniform_fr(1 << 14, 1001), threads=4)
    assert bytes.fromhex(m26["result_hex"]) == want
    # pianist object: worker rows 0 and 1 of a 2-machine setup, one per rank, and their aggregated commitment
    pk = rec["pianist_kzg22"]
    assert pk["n_gpus"] == 2 and len(pk["results_hex_by_rank"]) == 2
    alpha = uniform_fr(1, 1)
    comms = []
    for r in range(2):
        e.gen_srs(TAU, 0, 12, 1, factors=[lagrange_factor(r, 1, (TAU * 7 + 1) % R_MOD)])
        srs = e.srs_read(0, 1 << 11)
        row = uniform_fr(1 << 11, r)
        want = oc.commit(srs, row, True) + b"".join(oc.open_(srs, row, alpha, True))
        assert bytes.fromhex(pk["results_hex_by_rank"][r]) == want, r
        comms.append(want[:48])
    assert bytes.fromhex(pk["aggregate_commitment_hex"]) == e.g1_sum_compressed(b"".join(comms))


def test_bench_multi_rank_failures_never_cost_the_headline(hip):
    """VERDICT r4 task 1: one rank's failure in an EXTRA workload must neither park the other ranks in a collective nor
    lose the headline.  Two ranks on this one GPU (gloo control plane; the library's RCCL preflight fails on a shared
    device and falls back), with a failure injected on rank 1
      * while it builds the msm26 tables (an OOM would look like this): every rank learns of it through the store before
        anybody enters a collective -> rc 0, headline, pianist_kzg22 measured, msm26 = {"error": "setup failed", ...};
      * in the MIDDLE of msm26's timed loop: rank 0 is then alone in an all_gather, which times out (process-group timeout,
        15 s here) -> rc 0, headline, msm26 = {"error": "timed region failed", ...}, the group marked unusable.
    And the parent-side watchdog of the launcher-less form terminates a launch that overruns it, with a non-zero code."""
    import subprocess
    import sys

    base = dict(os.environ, BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", BENCH_COMM_INIT_TIMEOUT_S="60", BENCH_PG_TIMEOUT_S="15")
    base.pop("BENCH_BACKEND", None)
    args = ["--gpus", "2", "--log-n", "13", "--steps", "4", "--warmup", "1", "--msm26-log", "15", "--kzg22-log", "11",
            "--no-cpu-baseline", "--no-pipelined"]

    def launch(fault, port):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(base, BENCH_FAULT=fault))
        assert out.returncode == 0, (fault, out.stderr[-3000:])
        jl = [json.loads(ln) for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(jl) == 2 and "partial_line" in jl[0] and jl[0]["value"] == jl[1]["value"] > 0
        return jl[-1]

    rec = launch("msm26_setup:1", 29571)
    assert rec["msm26"]["error"] == "setup failed" and list(rec["msm26"]["ranks"]) == ["1"] and "injected fault" in rec["msm26"]["ranks"]["1"]
    assert rec["pianist_kzg22"]["value"] > 0 and len(rec["pianist_kzg22"]["results_hex_by_rank"]) == 2
    assert "process_group_note" not in rec
    rec = launch("msm26_step:1", 29572)
    assert rec["msm26"]["error"] == "timed region failed" and set(rec["msm26"]["ranks"]) == {"0", "1"}
    assert "injected fault" in rec["msm26"]["ranks"]["1"] and rec["pianist_kzg22"]["value"] > 0
    assert "msm26 failed inside its timed region" in rec["process_group_note"]
    rec = launch("pianist_kzg22_setup:0", 29573)
    assert rec["pianist_kzg22"]["error"] == "setup failed" and rec["msm26"]["value"] > 0 and rec["msm26"]["all_ranks_equal"]
    # the watchdog of `python bench.py --gpus 2` (no launcher): half a second is not enough for two ranks to even import torch
    env = {k: v for k, v in dict(base, BENCH_WATCHDOG_S="0.5").items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=env)
    assert out.returncode not in (0, 2) and "terminating it" in out.stderr and "killed by the watchdog" in out.stderr


# ------------------------------------------------------------------ seeded fuzz slice + the reference's fault scenarios
def test_seeded_fuzz_slice():
    """A fixed-seed slice of tests/fuzz_gpu.py inside the driver's `pytest -m gpu` run (the hours of fuzzing under
    profiles/*_fuzz_*.log are builder-side evidence only): 14 engine rounds of random size / window / slice -- every MSM
    through the blocking, resident and ticketed entry points on uniform / small / edge / equal / clustered scalars, NTT
    against the oracle and its round trip, commit / open / fused commit+open incl. alpha = 0, 1, omega^k, r - 1, the text
    path with row-cache hits and one-coefficient mutations, the fused transform + evaluation.  A mismatch raises inside
    run() naming its case; the floors make sure every kind really ran."""
    from tests import fuzz_gpu

    stats = fuzz_gpu.run(budget=600.0, seed=20261201, rounds=14, max_log=17)
    assert stats["rounds"] == 14 and stats["msm"] == 42 and stats["ntt"] == 14 and stats["kzg"] == 28
    assert stats["cache_hits"] >= 6 and stats["cache_misses_after_mutation"] >= 6
    assert stats["cache_hits"] + stats["cache_misses_after_mutation"] == 28
    assert stats["fused_from_text"] == 28 and stats["fused_from_text_streamed"] >= 4


@pytest.mark.parametrize("missing_info,too_late,invalid_proof,half_time,expected",
                         [(False, False, False, False, [1.0, 1.0]), (True, False, False, False, [0.0, 1.0]),
                          (False, True, False, False, [0.0, 1.0]), (False, False, True, False, [0.0, 1.0]),
                          (False, False, False, True, [0.5, 1.0])])
def test_reference_reward_scenarios_on_the_hip_engine(missing_info, too_late, invalid_proof, half_time, expected):
    """The reference's whole notion of fault injection (tests/test_validator.py:60-121), scenario for scenario, with
    every proof produced by the HIP engine and every check a real pairing: ok / commitment missing / answer late /
    proof + 1 as a big-endian integer / half the timeout used.  timeout = 10 s as in the reference's test."""
    from zkp_subnet_amd.client import Client
    from zkp_subnet_amd.miner import Miner, default_config
    from zkp_subnet_amd.validator import generate_challenge, reward

    client = Client(seed=31)
    client.start(scale=6, machines_scale=2)
    miner = Miner(default_config(scale=6, machines_scale=2, seed=31), client=client)
    try:
        ch = generate_challenge(client, 2)
        responses = [miner.forward(ch.to_synapse(i)) for i in range(2)]
        times = [0.0, 0.0]
        timeout = 10.0
        if missing_info:
            responses[0] = responses[0].model_copy(update={"commitment": None})
        if too_late:
            times[0] = 11.0
        if invalid_proof:
            raw = base64.b64decode(responses[0].proof)
            bumped = (int.from_bytes(raw, "big") + 1) % (1 << (8 * len(raw)))
            responses[0] = responses[0].model_copy(update={"proof": base64.b64encode(bumped.to_bytes(len(raw), "big")).decode()})
        if half_time:
            times[0] = 5.0
        got = [reward(client, ch, responses[i], i, times[i], timeout) for i in range(2)]
        assert got == expected
        from zkp_subnet_amd.validator import get_rewards        # the reference's array form (neurons/validator.py:178-192)
        assert [float(x) for x in get_rewards(client, ch, responses, times, timeout)] == expected
    finally:
        miner.stop()


def test_lane_machinery_stress_eight_host_threads_one_context(hip, tmp_path):
    """The reference's axon runs Miner.forward on worker threads and must never take the process down
    (neurons/miner.py:106-135).  Eight host threads drive ONE context for ~20 s with a random mix of everything that
    touches the lane machinery: worker_commit / worker_open through the row cache (hits, and misses after a one-coefficient
    mutation), the fused call, plain and resident MSMs, tickets that are waited for and tickets that are cancelled,
    kzg_upload_fr, and SRS reloads from a setup file -- one that fails (a point off the curve: the old tables must keep
    serving) and one that succeeds (same points).  EVERY answer is compared with the oracle's; the only failures allowed
    are the documented KZG_E_BUSY cases (include/kzg_mi355x.h: every lane parked under tickets, or a whole-context
    operation while a ticket is out); no thread may hang."""
    import threading
    import time

    from zkp_subnet_amd import codec
    from zkp_subnet_amd._native import KZG_E_BUSY, KZG_E_POINT, KzgError

    lg, ms = 11, 1
    T = 1 << (lg - ms)
    tx, ty = 0x5EED0001, 0x5EED0002
    eng = hip()
    eng.gen_srs(tx, ty, lg, ms)                                  # both slices resident: 2 x 2^10 points
    flat = eng.srs_read(0, 2 * T)
    good_file, bad_file = str(tmp_path / "setup_ok.uncompressed"), str(tmp_path / "setup_bad.uncompressed")
    with open(good_file, "wb") as f:
        f.write(flat)
    broken = bytearray(flat)
    broken[96 * 777 + 95] ^= 1                                   # y of point 777 leaves the curve
    with open(bad_file, "wb") as f:
        f.write(bytes(broken))
    srs = [flat[:96 * T], flat[96 * T:]]
    assert srs[0] == oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), lg, ms, 0)
    # ---- the answer book (oracle only)
    K = 4
    rows = [rand_scalars_bytes(T, 900 + k) for k in range(K)]
    alphas = [rand_scalars_bytes(1, 950 + k) for k in range(K)]
    polys = [codec.be32_to_fr_list(r) for r in rows]
    mut_rows, mut_polys = [], []
    for k in range(K):                                           # the same row with ONE coefficient changed
        j = 37 * (k + 1)
        v = (int.from_bytes(rows[k][32 * j:32 * j + 32], "big") + 1) % o.R
        mr = rows[k][:32 * j] + v.to_bytes(32, "big") + rows[k][32 * j + 32:]
        mut_rows.append(mr)
        mut_polys.append(codec.be32_to_fr_list(mr))
    want_c = {(w, k): oc.commit(srs[w], rows[k], True) for w in range(2) for k in range(K)}
    want_o = {(w, k): oc.open_(srs[w], rows[k], alphas[k], True) for w in range(2) for k in range(K)}
    want_om = {(w, k): oc.open_(srs[w], mut_rows[k], alphas[k], True) for w in range(2) for k in range(K)}
    slots = [rand_scalars_bytes(2 * T, 980), rand_scalars_bytes(T, 981)]       # resident scalar sets of slots 0 and 1
    msm_cases = [(0, 2 * T, 0), (0, T, T), (0, 300, 123), (1, T, 0), (1, 512, 1024)]
    want_m = {c: oc.msm(flat[96 * c[2]:96 * (c[2] + c[1])], slots[c[0]][:32 * c[1]]) for c in msm_cases}
    for s, data in enumerate(slots):
        eng.upload_fr(s, data, False)
    # ---- the hammer
    deadline = time.time() + 20.0
    errors, counts, busy = [], {}, {}
    lock = threading.Lock()

    def note(table, key):
        with lock:
            table[key] = table.get(key, 0) + 1

    def worker(tid):
        rnd = random.Random(7000 + tid)
        ops = ["commit", "open_hit", "open_miss", "fused", "msm", "msm_res", "ticket", "cancel", "upload", "reload_bad",
               "reload_ok"]
        weights = [6, 6, 3, 6, 4, 6, 6, 3, 2, 1, 1]
        while time.time() < deadline and not errors:
            op = rnd.choices(ops, weights)[0]
            w, k = rnd.randrange(2), rnd.randrange(K)
            try:
                if op == "commit":
                    ok = eng.commit_list(w, polys[k], True) == want_c[(w, k)]
                elif op == "open_hit":                          # the unchanged miner's pair: the second call may hit
                    ok = eng.commit_list(w, polys[k], True) == want_c[(w, k)] and \
                        eng.open_list(w, polys[k], alphas[k], True) == want_o[(w, k)]
                elif op == "open_miss":                         # commit one row, open its mutation: never the cached answer
                    ok = eng.commit_list(w, polys[k], True) == want_c[(w, k)] and \
                        eng.open_list(w, mut_polys[k], alphas[k], True) == want_om[(w, k)]
                elif op == "fused":
                    ok = eng.commit_open(w, rows[k], alphas[k], True) == (want_c[(w, k)],) + want_o[(w, k)]
                elif op == "msm":
                    c = rnd.choice(msm_cases)
                    ok = eng.msm(slots[c[0]][:32 * c[1]], c[2]) == want_m[c]
                elif op == "msm_res":
                    c = rnd.choice(msm_cases)
                    ok = eng.msm_resident(*c) == want_m[c]
                elif op == "ticket":
                    c = rnd.choice(msm_cases)
                    part = rnd.random() < 0.5
                    t = eng.msm_submit(c[0], c[1], c[2], partial=part)
                    if rnd.random() < 0.5:
                        time.sleep(rnd.random() * 0.002)        # others run into the parked lane meanwhile
                    r = eng.msm_wait(t)
                    ok = (eng.g1_sum(r) if part else r) == want_m[c]
                elif op == "cancel":
                    c = rnd.choice(msm_cases)
                    t = eng.msm_submit(c[0], c[1], c[2])
                    eng.msm_cancel(t)
                    ok = True
                elif op == "upload":                            # exclusive; same content, so the answer book holds
                    s = rnd.randrange(2)
                    eng.upload_fr(s, slots[s], False)
                    ok = True
                elif op == "reload_bad":
                    try:
                        eng.load_srs_file(bad_file, lg, ms)
                        ok = False                              # must not load
                    except KzgError as e:
                        if e.code == KZG_E_BUSY:
                            raise
                        ok = e.code == KZG_E_POINT
                else:
                    eng.load_srs_file(good_file, lg, ms)
                    ok = True
                if not ok:
                    errors.append((tid, op, w, k, "wrong answer"))
                note(counts, op)
            except KzgError as e:
                if e.code == KZG_E_BUSY:
                    note(busy, op)
                else:
                    errors.append((tid, op, w, k, repr(e)))
            except Exception as e:                              # noqa: BLE001
                errors.append((tid, op, w, k, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=max(1.0, deadline + 90.0 - time.time()))
    hung = [i for i, t in enumerate(threads) if t.is_alive()]
    assert not hung, f"threads {hung} never came back (deadlock in the lane machinery?) counts={counts} busy={busy}"
    assert not errors, (errors[:5], counts, busy)
    done = sum(counts.values())
    assert done >= 200 and all(counts.get(op, 0) > 0 for op in ("commit", "open_hit", "open_miss", "fused", "msm", "msm_res",
                                                                "ticket", "cancel", "upload", "reload_bad", "reload_ok")), (counts, busy)
    assert sum(busy.values()) < done, (counts, busy)             # E_BUSY is the exception, not the rule
    # after the storm: the context still answers, tickets all returned, results unchanged
    assert eng.commit_open(1, rows[0], alphas[0], True) == (want_c[(1, 0)],) + want_o[(1, 0)]
    assert eng.msm_resident(*msm_cases[0]) == want_m[msm_cases[0]]
    hits, misses = eng.row_cache_stats()
    assert hits > 0 and misses > 0
    print("lane stress:", {"done": counts, "busy": busy, "cache": (hits, misses)})


def test_multi_device_client_two_contexts_on_one_gpu():
    """MultiDeviceClient with G = 2 contexts (both on device 0: this box has one GPU; on a multi-GPU host the list names
    different devices): worker index i is served by context i mod 2, each context generated only its own slices, the four
    rows of a challenge run concurrently from host threads, and every commitment / evaluation / proof equals the C
    oracle's on the same slice."""
    from zkp_subnet_amd import MultiDeviceClient, codec
    from zkp_subnet_amd.client import derive_taus
    from zkp_subnet_amd.validator import generate_challenge, verify_all

    lg, ms = 12, 2
    multi = MultiDeviceClient(devices=[0, 0], seed=77)
    multi.start(scale=lg, machines_scale=ms)
    try:
        assert [c.workers for c in multi.clients] == [[0, 2], [1, 3]]
        ch = generate_challenge(multi, 4)
        answers = multi.commit_and_open_rows(range(4), ch.polys, ch.alpha)
        tx, ty = (t.to_bytes(32, "big") for t in derive_taus(77))
        alpha = codec.fr_to_be32(ch.alpha)
        for i, a in enumerate(answers):
            assert a.status_code == 200, a.json()
            srs = oc.srs_gen(tx, ty, lg, ms, i)
            row = codec.fr_list_to_be32(ch.polys[i])
            ev, pf = oc.open_(srs, row, alpha, True)
            assert codec.g1_from_b64(a.json()["commitment"]) == oc.commit(srs, row, True), i
            assert (codec.fr_to_be32(a.json()["eval"]), codec.g1_from_b64(a.json()["proof"])) == (ev, pf), i
            assert a.json()["eval"] == ch.evals[i]                 # the validator's own evaluation, from the other path
        from zkp_subnet_amd.protocol import Prove
        responses = [Prove(index=i, poly=[], commitment=a.json()["commitment"], proof=a.json()["proof"], eval=a.json()["eval"])
                     for i, a in enumerate(answers)]
        assert verify_all(multi, ch, responses, threads=4) == [True] * 4
    finally:
        multi.stop()


def test_row_cache_hit_is_verified_against_the_row_not_trusted_to_the_tag(hip):
    """kzg_commit_cached / kzg_open_cached take a 128-bit content tag from the caller (the codec's keyed hash: fast, but
    with no cryptographic analysis -- ADVICE r3).  The tag is only a hint: on a hit the caller's row is uploaded beside
    the request and compared bit for bit with the row the slot was filled from.  TWO DIFFERENT rows under the SAME tag
    must each get their own, oracle-equal answers -- never the other row's proof -- and the colliding slot is dropped."""
    import ctypes

    eng = hip()
    lg = 10
    T = 1 << lg
    eng.gen_srs(0xC0111DE, 1, lg, 0)
    srs = eng.srs_read(0, T)
    row_a, row_b = rand_scalars_bytes(T, 1201), rand_scalars_bytes(T, 1202)
    row_b2 = row_a[:32 * 500] + row_b[32 * 500:32 * 501] + row_a[32 * 501:]       # differs from A in ONE element
    alpha = rand_scalars_bytes(1, 1203)
    tag = bytes(range(16))

    def commit(row):
        out = ctypes.create_string_buffer(48)
        eng._chk(eng._lib.kzg_commit_cached(eng._h, 0, row, T, 1, tag, out))
        return out.raw

    def open_(row):
        ev, pf = ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
        eng._chk(eng._lib.kzg_open_cached(eng._h, 0, row, T, 1, tag, alpha, ev, pf))
        return ev.raw, pf.raw

    h0, m0 = eng.row_cache_stats()
    assert commit(row_a) == oc.commit(srs, row_a, True)                        # miss: fills the slot under `tag`
    assert open_(row_a) == oc.open_(srs, row_a, alpha, True)                   # genuine hit
    assert eng.row_cache_stats() == (h0 + 1, m0 + 1)
    for other in (row_b, row_b2):
        assert open_(other) == oc.open_(srs, other, alpha, True)               # same tag, different row: recomputed
        assert commit(other) == oc.commit(srs, other, True)
        assert open_(row_a) == oc.open_(srs, row_a, alpha, True)               # ... and A is still A
    # accounting: round 1 -- open(B) collides (counted as a miss, slot dropped), commit(B) misses and fills the slot,
    # open(A) collides with it; round 2 -- open(B2) misses and fills, commit(B2) is a GENUINE hit, open(A) collides
    assert eng.row_cache_stats() == (h0 + 2, m0 + 6)                           # no collision was ever counted as a hit


def test_calibrate_reports_a_plausible_mad_rate(hip):
    """kzg_calibrate (what bench.py's mad_issue.peak comes from): the v_mad_u64_u32 issue rate of this GPU, measured on
    the spot.  Plausibility only -- 1.5 .. 4 ns per wave-instruction per SIMD at >= 2 waves per SIMD (2.30 - 2.36 measured
    on this pool), about twice that for a lone wave, 1024 SIMDs, a clock between 1 and 3 GHz -- and the documented
    failures: bad argument, and KZG_E_BUSY while a ticket is out (it needs the whole context)."""
    from zkp_subnet_amd._native import KZG_E_ARG, KZG_E_BUSY, KzgError

    eng = hip()
    two, one = eng.calibrate(2), eng.calibrate(1)
    assert two["simds"] == 1024 and 1.5 < two["ns_per_mad_per_simd"] < 4.0, two
    assert abs(two["gmad_per_s"] - 1024 / two["ns_per_mad_per_simd"]) < 1e-6
    assert 1.4 * two["ns_per_mad_per_simd"] < one["ns_per_mad_per_simd"] < 3.0 * two["ns_per_mad_per_simd"], (one, two)
    assert 1.0 < two["memtime_ticks_per_ns"] < 3.0 and 0.5 < two["kernel_ms"] < 10.0
    with pytest.raises(KzgError) as ei:
        eng.calibrate(9)
    assert ei.value.code == KZG_E_ARG
    eng.gen_srs(0x77, 1, 8, 0)
    eng.upload_fr(0, rand_scalars_bytes(256, 5), False)
    t = eng.msm_submit(0, 256, 0)
    with pytest.raises(KzgError) as ei:
        eng.calibrate(2)
    assert ei.value.code == KZG_E_BUSY
    got = eng.msm_wait(t)
    assert got == eng.msm_resident(0, 256, 0)
    assert 1.5 < eng.calibrate(2)["ns_per_mad_per_simd"] < 4.0


def test_tile_streamed_upload_of_long_rows_matches_the_one_shot_path(hip, monkeypatch):
    """Long rows are decoded tile by tile, each tile's upload started at once (kzg_staging_flush) so that the copy engine
    works while the codec decodes; the compute call then finds the row on the device.  Forced here at 2^12 in four tiles:
    fused call, the two-call route (miss, verified hit -- the verification reads the flushed twin --, mutated row) all
    equal the oracle; and the flush entry point refuses anything but the next contiguous piece."""
    import ctypes

    from zkp_subnet_amd import HipEngine, codec
    from zkp_subnet_amd._native import KZG_E_ARG, KzgError

    monkeypatch.setattr(HipEngine, "STREAM_MIN", 1 << 12)
    monkeypatch.setattr(HipEngine, "STREAM_TILE", 1 << 10)
    eng = hip()
    lg = 12
    T = 1 << lg
    eng.gen_srs(0x57AEA3, 1, lg, 0)
    srs = eng.srs_read(0, T)
    row = rand_scalars_bytes(T, 1301)
    alpha = rand_scalars_bytes(1, 1302)
    poly = codec.be32_to_fr_list(row)
    c = oc.commit(srs, row, True)
    ev, pf = oc.open_(srs, row, alpha, True)
    assert eng.commit_open_list(0, poly, alpha, True) == (c, ev, pf)
    h0, m0 = eng.row_cache_stats()
    assert eng.commit_list(0, poly, True) == c
    assert eng.open_list(0, poly, alpha, True) == (ev, pf)
    assert eng.row_cache_stats() == (h0 + 1, m0 + 1)
    row2 = row[:32 * 3000] + (5).to_bytes(32, "big") + row[32 * 3001:]
    assert eng.open_list(0, codec.be32_to_fr_list(row2), alpha, True) == oc.open_(srs, row2, alpha, True)
    # a coefficient-form row and a shorter one through the same path (T < STREAM_MIN: one shot)
    assert eng.commit_open_list(0, poly, alpha, False) == (oc.commit(srs, row, False),) + oc.open_(srs, row, alpha, False)
    assert eng.commit_open_list(0, poly[:1024], alpha, False) == (oc.commit(srs, row[:32 * 1024], False),) + oc.open_(srs, row[:32 * 1024], alpha, False)
    # the entry point itself
    ptr, tok = ctypes.c_void_p(), ctypes.c_int(-1)
    eng._chk(eng._lib.kzg_staging_acquire(eng._h, 32 * T, ctypes.byref(ptr), ctypes.byref(tok)))
    ctypes.memmove(ptr.value, row, len(row))
    for args in ((32 * 1024, 32 * 1024), (0, 33), (0, 1 << 40)):             # not at the flushed prefix / ragged / beyond the buffer
        with pytest.raises(KzgError) as ei:
            eng._chk(eng._lib.kzg_staging_flush(eng._h, tok.value, *args))
        assert ei.value.code == KZG_E_ARG
    eng._chk(eng._lib.kzg_staging_flush(eng._h, tok.value, 0, 32 * 2048))     # HALF the row flushed: the call uploads it all itself
    out = ctypes.create_string_buffer(48)
    eng._chk(eng._lib.kzg_commit(eng._h, 0, ctypes.cast(ptr, ctypes.c_char_p), T, 1, out))
    assert out.raw == c
    eng._chk(eng._lib.kzg_staging_flush(eng._h, tok.value, 32 * 2048, 32 * 2048))   # ... now all of it: served from the twin
    eng._chk(eng._lib.kzg_commit(eng._h, 0, ctypes.cast(ptr, ctypes.c_char_p), T, 1, out))
    assert out.raw == c
    # the flushes are ONE-SHOT: the holder rewrites the pinned buffer and calls again without releasing --
    # the call must answer for the NEW bytes (ordinary upload), never for the stale twin
    ctypes.memmove(ptr.value, row2, len(row2))
    c2 = oc.commit(srs, row2, True)
    assert c2 != c
    eng._chk(eng._lib.kzg_commit(eng._h, 0, ctypes.cast(ptr, ctypes.c_char_p), T, 1, out))
    assert out.raw == c2
    with pytest.raises(KzgError):                                             # ... and the next flush starts over from 0
        eng._chk(eng._lib.kzg_staging_flush(eng._h, tok.value, 32 * 4096, 0))
    eng._chk(eng._lib.kzg_staging_flush(eng._h, tok.value, 0, 32 * 4096))
    eng._chk(eng._lib.kzg_commit(eng._h, 0, ctypes.cast(ptr, ctypes.c_char_p), T, 1, out))
    assert out.raw == c2
    with pytest.raises(KzgError):                                             # offset + bytes must not wrap
        eng._chk(eng._lib.kzg_staging_flush(eng._h, tok.value, 0, (1 << 64) - 32))
    eng._chk(eng._lib.kzg_staging_release(eng._h, tok.value))
    with pytest.raises(KzgError):
        eng._chk(eng._lib.kzg_staging_flush(eng._h, tok.value, 0, 32))        # not held any more
