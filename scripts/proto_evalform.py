"""Timing prototype for the evaluation-form (Lagrange-table) path of VERDICT r3 task 5 -- what it COULD buy, measured
with the kernels that exist, before anything is built (dev tool; needs the GPU; results of the stand-in calls are NOT
group elements anybody wants: only their timing is used).

Part A -- the ceiling.  A commitment over a Lagrange table is MSM(table, the row as received): no INTT.  The existing
entry points already run exactly that sequence when told the row is in coefficient form (evaluation_form = 0: the same
decode, sort, accumulate, tree over a table of the same shape; only the table's CONTENT differs).  An evaluation-form
opening (barycentric y, pointwise quotient) still has to pass over the row and needs a batched inversion on top, so
"commit+open with evaluation_form = 0" is what a full evaluation-form path would cost if its opening arithmetic were as
cheap as Horner + synthetic division and its inversion free: an UPPER bound on the gain.

Part B -- what the table would unlock at long rows: the commitment's MSM consuming the row tile by tile while the text
codec is still decoding.  Emulated with existing calls: a decoder thread fills one pinned staging buffer tile by tile, a
second thread runs kzg_msm_partial on each finished tile (the table's content does not change its timing), then
kzg_open on the whole row (INTT + quotient + MSM as today) and the sum of the tile partials.  The emulation uploads the
row twice (tile by tile for the commitment, whole for kzg_open); the redundant upload is measured and reported.

    python scripts/proto_evalform.py [log2_T ...]"""
import ctypes
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import TAU, R_MOD, uniform_fr                       # noqa: E402
from zkp_subnet_amd import HipEngine, codec                    # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [12, 16, 20, 22]


def med(f, reps, warm=2):
    for _ in range(warm):
        f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    return t[len(t) // 2]


for lg in sizes:
    T = 1 << lg
    eng = HipEngine(0)
    eng.gen_srs(TAU, (TAU * 7 + 1) % R_MOD, lg, 0)
    row = uniform_fr(T, 1)
    alpha = uniform_fr(1, 2)
    poly = codec.be32_to_fr_list(row)
    reps = 40 if lg <= 16 else 6
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.1:
        eng.commit_open(0, row, alpha, True)
    out = {"log2_T": lg, "window_bits": eng.window}
    # ---- A: evaluation_form = 1 (today: INTT first) against = 0 (the sequence a Lagrange-table path would run)
    eng.upload_fr(0, row, True)
    for ef in (1, 0):
        k = "today_ef1" if ef else "no_intt_ef0"
        out[f"resident_commit_open_ms_{k}"] = round(med(lambda: eng.commit_open_resident(0, 0, T, alpha, bool(ef)), reps), 4)
        out[f"host_commit_ms_{k}"] = round(med(lambda: eng.commit(0, row, bool(ef)), reps), 4)
        out[f"host_open_ms_{k}"] = round(med(lambda: eng.open(0, row, alpha, bool(ef)), reps), 4)
        out[f"text_fused_ms_{k}"] = round(med(lambda: eng.commit_open_list(0, poly, alpha, bool(ef)), reps), 4)

        def two_call():
            eng.commit_list(0, poly, bool(ef))
            eng.open_list(0, poly, alpha, bool(ef))
        out[f"text_two_call_ms_{k}"] = round(med(two_call, reps), 4)
    for key in ("resident_commit_open_ms", "host_commit_ms", "host_open_ms", "text_fused_ms", "text_two_call_ms"):
        a, b = out[f"{key}_today_ef1"], out[f"{key}_no_intt_ef0"]
        out[f"{key}_ceiling_gain_pct"] = round((a - b) / a * 100, 2)
    # ---- B: tile-streamed commitment (long rows only)
    if lg >= 20:
        for tile_lg in (18, 19, 20):
            if tile_lg >= lg:
                continue
            tile = 1 << tile_lg
            ntiles = T // tile
            tiles = [poly[k * tile:(k + 1) * tile] for k in range(ntiles)]       # pre-split: a C-level range decode is free
            ptr, tok = ctypes.c_void_p(), ctypes.c_int(-1)
            eng._chk(eng._lib.kzg_staging_acquire(eng._h, 32 * T, ctypes.byref(ptr), ctypes.byref(tok)))
            base = ptr.value

            def streamed():
                ready = [threading.Event() for _ in range(ntiles)]
                parts = [None] * ntiles

                def decoder():
                    for k in range(ntiles):
                        codec._wire.decode_fr_list_into(tiles[k], base + 32 * k * tile, 32 * tile)
                        ready[k].set()

                def msm():
                    for k in range(ntiles):
                        ready[k].wait()
                        o = ctypes.create_string_buffer(192)
                        eng._chk(eng._lib.kzg_msm_partial(eng._h, ctypes.c_char_p(base + 32 * k * tile), tile, k * tile, o))
                        parts[k] = o.raw
                td, tm = threading.Thread(target=decoder), threading.Thread(target=msm)
                td.start(); tm.start(); td.join(); tm.join()
                ev, pf = ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
                eng._chk(eng._lib.kzg_open(eng._h, 0, ctypes.c_char_p(base), T, 1, alpha, ev, pf))
                return eng.g1_sum(b"".join(parts)), ev.raw, pf.raw
            out[f"text_streamed_tiles_2^{tile_lg}_ms"] = round(med(streamed, 5, 1), 3)
            eng._lib.kzg_staging_release(eng._h, tok.value)
        # the redundant second upload inside the emulation: open from the host buffer against open of a resident row
        out["host_open_ms"] = out["host_open_ms_today_ef1"]
        out["redundant_upload_ms_estimate"] = round(
            out["host_commit_ms_today_ef1"] + out["host_open_ms_today_ef1"] - out["resident_commit_open_ms_today_ef1"]
            - 0.0, 3)      # two uploads + two decodes: the fused call pays one
        best = min(v for k, v in out.items() if k.startswith("text_streamed"))
        out["text_streamed_best_ms"] = best
        out["text_fused_today_ms"] = out["text_fused_ms_today_ef1"]
    print(json.dumps(out), flush=True)
    eng.close()
