"""GPU parity tests (`-m gpu`), kzg: KZG commit / open of worker rows: golden vectors, oracle parity at every path (batched, two-lane, cached two-call route, tile-streamed upload), Pianist rows and their aggregation.
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


# rows up to 2^18 take the batched two-set pass, longer ones the two-lane form (pipeline.hip commit_open_dev): both sides of
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
