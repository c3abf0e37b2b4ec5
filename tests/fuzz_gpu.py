"""Randomised cross-check of the HIP path against the C oracle (not collected by pytest; needs the GPU; lives under
tests/ because it uses the oracle):
MSM (random size / window / scalar distribution / offset), NTT round trips and oracle equality, commit / open /
commit+open on random rows incl. special alphas, the same through the text path and the row cache (with and without a
coefficient changed between the two calls), the fused transform + evaluation, and ONE MSM cut over the G contexts of one
handle (kzg_multi_msm: random G, size, range and scalar distribution against the single-context MSM and the oracle).
`python tests/fuzz_gpu.py [seconds] [seed]`; a seeded slice (`run(rounds=...)`) is part of the driver's `pytest -m gpu` run
(tests/test_gpu_bench.py::test_seeded_fuzz_slice)."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bls12_381 as o          # noqa: E402
from oracle import cpu as oc               # noqa: E402
from zkp_subnet_amd import HipEngine, SegmentedMsm, codec  # noqa: E402



def run(budget=60.0, seed=None, rounds=None, max_log=20, with_comm=False):
    """Fuzz for `budget` seconds, or -- with `rounds` -- for exactly that many engine rounds (a fixed seed then gives the
    same cases on every box); any mismatch raises AssertionError naming the case.  Returns the case counts by kind.
    `with_comm` (the long builder-side runs): every fourth round also drives the library's own collective on a one-rank
    communicator (kzg_comm_init / kzg_msm_sharded) against the same oracle answers.  A progress line goes out every two
    minutes, so that a run cut short by its lease still says how far it got (round 4's hour-long run left only its seed)."""
    seed = int(time.time()) if seed is None else seed
    rnd = random.Random(seed)
    print("seed", seed, flush=True)
    oc.build()
    t_start = time.time()
    t_end = t_start + budget
    t_note = t_start + 120.0
    stats = {"rounds": 0, "msm": 0, "ntt": 0, "kzg": 0, "cache_hits": 0, "cache_misses_after_mutation": 0, "sharded": 0,
             "segmented": 0}

    def scalars(n, kind):
        if kind == "uniform":
            return b"".join(rnd.randrange(o.R).to_bytes(32, "big") for _ in range(n))
        if kind == "small":
            return b"".join(rnd.randrange(1 << rnd.choice((1, 8, 33, 64))).to_bytes(32, "big") for _ in range(n))
        if kind == "edge":
            pool = [0, 1, 2, o.R - 1, o.R - 2, (o.R - 1) // 2, (o.R + 1) // 2, (1 << 254), (1 << 255) % o.R] + \
                   [(1 << k) - 1 for k in (8, 16, 20, 28, 32, 64, 128, 200)] + [(1 << k) for k in (7, 15, 19, 27, 31, 63, 127)]
            return b"".join(rnd.choice(pool).to_bytes(32, "big") for _ in range(n))
        if kind == "equal":
            return rnd.randrange(1, o.R).to_bytes(32, "big") * n
        if kind == "clustered":   # low digits spread over a few adjacent buckets: oversized sort partitions without a dominant bucket
            base = rnd.randrange(o.R >> 1) & ~((1 << 40) - 1)
            spread = rnd.choice((4, 6, 9, 12))
            return b"".join((base + rnd.randrange(1 << spread)).to_bytes(32, "big") for _ in range(n))
        few = [rnd.randrange(o.R) for _ in range(3)]
        return b"".join(rnd.choice(few).to_bytes(32, "big") for _ in range(n))


    while (stats["rounds"] < rounds) if rounds else (time.time() < t_end):
        # ---- one engine / SRS per round
        lg = rnd.choice((3, 5, 8, 10, 11, 12, 13, 14, 15, 16, 16, 17, 18)) if rnd.random() < 0.93 else rnd.choice((19, 20))
        lg = min(lg, max_log)
        ms = rnd.choice((0, 0, 1, 2)) if lg >= 4 else 0
        window = rnd.choice((0, 0, 0, 4, 5, 7, 8, 9, 11, 12, 13, 15, 16, 17, 18))
        if lg <= 14 and rnd.random() < 0.15:
            window = rnd.choice((19, 20, 22, 24))       # the widest windows (what 2^20 .. 2^26 slices use) on small inputs
        tx, ty = rnd.randrange(2, o.R), rnd.randrange(2, o.R)
        eng = HipEngine(0, window=window)
        i = rnd.randrange(1 << ms)
        eng.gen_srs(tx, ty, lg, ms, [i])
        T = 1 << (lg - ms)
        srs = oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), lg, ms, i)
        assert eng.srs_read(0, T) == srs, ("srs", lg, ms, i)
        comm = with_comm and stats["rounds"] % 4 == 0
        if comm:
            eng.comm_init(HipEngine.comm_unique_id(), 0, 1, timeout_ms=60000)
            eng.comm_selftest()
        for _ in range(3):
            n = rnd.choice((T, T, rnd.randrange(1, T + 1)))
            off = rnd.randrange(0, T - n + 1)
            kind = rnd.choice(("uniform", "uniform", "small", "edge", "equal", "few", "clustered"))
            sc = scalars(n, kind)
            want = oc.msm(srs[96 * off:96 * (off + n)], sc)
            assert eng.msm(sc, off) == want, ("msm", lg, ms, window, n, off, kind)
            eng.upload_fr(1, sc, False)
            assert eng.msm_resident(1, n, off) == want
            t1, t2 = eng.msm_submit(1, n, off), eng.msm_submit(1, n, off, partial=True)
            assert eng.msm_wait(t1) == want and eng.g1_sum(eng.msm_wait(t2)) == want, ("ticket", lg, window, n)
            if comm:
                assert eng.msm_sharded(1, n, off) == want, ("sharded", lg, ms, window, n, off, kind)
                stats["sharded"] += 1
            stats["msm"] += 1
        # ---- NTT
        row = scalars(T, "uniform")
        f = eng.ntt(row, False)
        assert f == oc.fr_ntt(row, False) and eng.ntt(f, True) == row, ("ntt", lg - ms)
        stats["ntt"] += 1
        # ---- commit / open
        for _ in range(2):
            row = scalars(T, rnd.choice(("uniform", "uniform", "edge", "equal", "small")))
            ev_form = rnd.random() < 0.8
            omega = pow(7, (o.R - 1) // T, o.R) if T > 1 else 1
            alpha = rnd.choice((rnd.randrange(o.R), 0, 1, omega, pow(omega, rnd.randrange(T), o.R), o.R - 1)).to_bytes(32, "big")
            c = oc.commit(srs, row, ev_form, threads=8)
            ev, pf = oc.open_(srs, row, alpha, ev_form, threads=8)
            assert eng.commit_open(0, row, alpha, ev_form) == (c, ev, pf), ("commit_open", lg, ms, window, ev_form)
            assert eng.commit(0, row, ev_form) == c and eng.open(0, row, alpha, ev_form) == (ev, pf), ("commit/open", lg, ms)
            stats["kzg"] += 1
            # the unchanged miner's two calls from the wire text: the second is served from the row cache -- unless the row
            # changed in between, in which case it must be recomputed
            poly = codec.be32_to_fr_list(row)
            # the fused call from the wire text; for half of the rows of >= 2^10 elements in the tile-streamed form (decode
            # in four tiles, each tile's upload started at once) that long rows take by default
            stream = T >= 1024 and rnd.random() < 0.5
            saved = (HipEngine.STREAM_MIN, HipEngine.STREAM_TILE)
            if stream:
                HipEngine.STREAM_MIN, HipEngine.STREAM_TILE = T, T >> 2
            try:
                assert eng.commit_open_list(0, poly, alpha, ev_form) == (c, ev, pf), ("commit_open_list", lg, ms, stream)
            finally:
                HipEngine.STREAM_MIN, HipEngine.STREAM_TILE = saved
            stats["fused_from_text"] = stats.get("fused_from_text", 0) + 1
            stats["fused_from_text_streamed"] = stats.get("fused_from_text_streamed", 0) + int(stream)
            h0, m0 = eng.row_cache_stats()
            assert eng.commit_list(0, poly, ev_form) == c, ("commit_list", lg, ms)
            if rnd.random() < 0.5:
                assert eng.open_list(0, poly, alpha, ev_form) == (ev, pf), ("open_list hit", lg, ms)
                assert eng.row_cache_stats()[0] == h0 + 1
                stats["cache_hits"] += 1
            else:
                k = rnd.randrange(T)
                v = (int.from_bytes(row[32 * k:32 * k + 32], "big") + 1 + rnd.randrange(o.R - 1)) % o.R
                row2 = row[:32 * k] + v.to_bytes(32, "big") + row[32 * k + 32:]
                poly2 = list(poly)
                poly2[k] = codec.be32_to_fr(v.to_bytes(32, "big"))
                assert eng.open_list(0, poly2, alpha, ev_form) == oc.open_(srs, row2, alpha, ev_form, threads=8), ("open_list miss", lg, ms, k)
                assert eng.row_cache_stats()[0] == h0
                stats["cache_misses_after_mutation"] += 1
            if T > 1:
                inv = rnd.random() < 0.5
                assert eng.ntt_eval(row, inv, alpha) == oc.fr_eval(oc.fr_ntt(row, inv), alpha), ("ntt_eval", lg - ms, inv)
        # ---- one MSM over the G contexts of one handle (every third round; ms == 0 rounds only: the slice is then a flat SRS)
        if ms == 0 and stats["rounds"] % 3 == 0 and T >= 8:
            G = rnd.choice((2, 3, 4))
            npts = rnd.choice((T, rnd.randrange(G, T + 1)))
            seg = SegmentedMsm([0] * G)
            try:
                seg.gen_srs(tx, npts)                       # point j = [tau^j] G: the same points as this round's engine holds
                for _ in range(2):
                    n = rnd.randrange(1, npts + 1)
                    off = rnd.randrange(0, npts - n + 1)
                    sc = scalars(n, rnd.choice(("uniform", "small", "edge", "equal", "few")))
                    want = oc.msm(srs[96 * off:96 * (off + n)], sc)
                    assert seg.msm(sc, off) == want, ("segmented", lg, G, npts, n, off)
                    seg.upload(rnd.randrange(4), sc, off)
                    stats["segmented"] += 1
                sc = scalars(npts, "uniform")
                seg.upload(2, sc, 0)
                assert seg.msm_resident(2) == eng.msm(sc, 0), ("segmented resident", lg, G, npts)
            finally:
                seg.close()
        eng.close()
        stats["rounds"] += 1
        if time.time() > t_note:
            print(f"progress after {time.time() - t_start:.0f} s:", stats, flush=True)
            t_note = time.time() + 120.0
    print("fuzz ok", stats, flush=True)
    return stats


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else None, with_comm=True)
