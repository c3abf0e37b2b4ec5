"""GPU parity tests (`-m gpu`), msm: the Pippenger MSM pipeline: golden vectors, adversarial scalar distributions, BASELINE sizes through trapdoor identities, tickets.
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
