"""benchlib.extras -- the objects a bench line carries beside the headline: commit+open latency rows with their own roofline and
CPU baseline (`kzg_commit_open`), the reference's own route from text (`e2e_from_text`), and -- with a process group -- the other
multi-GPU configurations of BASELINE.json measured in the same launch (`pianist_kzg22`, `msm26`).  Every one runs AFTER the
headline's timed region and can never cost the headline line."""
import os
import sys
import time

from .common import *  # noqa: F401,F403
from .common import errmsg, host_cores, pctl, thread_counts, uniform_fr
from .control import inject

def kzg_rows_report(HipEngine, lagrange_factor, device, logs, cpu_threads, with_cpu):
    """commit+open latency of device-resident evaluation-form rows (N = 1): BASELINE.json configs[2] (2^22) and the row
    lengths the reference actually runs (mainnet 2^16, testnet 2^12: reference Makefile:63-116).  One engine per row
    length (the window tables are built for the slice length).  Never part of `value`."""
    rows = {}
    for lg in logs:
        T = 1 << lg
        eng = HipEngine(device)
        t0 = time.time()
        eng.gen_srs(TAU, (TAU * 7 + 1) % R_MOD, lg, 0)
        setup_s = time.time() - t0
        row = uniform_fr(T, seed=0)
        alpha = uniform_fr(1, seed=1)
        eng.upload_fr(0, row, True)
        warm, steps = (2, 8) if lg >= 20 else (5, 40)
        t_w = time.perf_counter()
        done = 0
        while done < warm or time.perf_counter() - t_w < 0.06:   # >= 60 ms of the same call: the clocks need ~40 ms of load
            ref = eng.commit_open_resident(0, 0, T, alpha, True)
            done += 1
        lat = []
        for _ in range(steps):
            t1 = time.perf_counter()
            got = eng.commit_open_resident(0, 0, T, alpha, True)
            lat.append((time.perf_counter() - t1) * 1e3)
            assert got == ref, "non-deterministic commit+open"
        eng.set_profiling(True)          # stage times: serialised on one lane so that they stay attributable
        stages = {}
        nprof = 3
        for _ in range(nprof):
            assert eng.commit_open_resident(0, 0, T, alpha, True) == ref
            for k, v in eng.timings().items():
                stages[k] = stages.get(k, 0.0) + v / nprof
        eng.set_profiling(False)
        med = pctl(lat, 0.5)
        alg = 384.0 * T                  # 64 INTT + 128 MSM + 64 quotient + 128 MSM bytes per coefficient
        ach = alg / (med * 1e-3) / 1e9
        rec = {"log2_T": lg, "window_bits": eng.window, "ms": round(med, 4), "p10": round(pctl(lat, 0.1), 4),
               "p90": round(pctl(lat, 0.9), 4), "steps": steps, "warmup_calls": done, "coefficients_per_s": T / (med * 1e-3),
               "stages_ms_profiled_serial": {k: round(v, 4) for k, v in stages.items()},
               "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes": alg, "per": "whole commit+open call"},
               "result_hex": b"".join(ref).hex(), "setup_s": round(setup_s, 2)}
        if with_cpu:
            from oracle import cpu as oc

            oc.build()
            m = min(T, 1 << 17)          # bounded CPU sample: a row of 2^17 coefficients at most
            srs = eng.srs_read(0, m)
            sample = row[: 32 * m]
            per = {}
            cpu_res = None
            for th in cpu_threads:
                if th == 1 and m > (1 << 14):
                    continue             # one thread on a long row would take minutes
                tc = time.perf_counter()
                c = oc.commit(srs, sample, True, threads=th)
                ev, pf = oc.open_(srs, sample, alpha, True, threads=th)
                per[th] = time.perf_counter() - tc
                cpu_res = (c, ev, pf)
            best = min(per, key=per.get)
            gpu_same = ref if m == T else eng.commit_open(0, sample, alpha, True)   # a shorter row on the same points
            rec["cpu_baseline"] = {
                "value": m / per[best], "unit": "coefficients/s", "cores": best, "kind": "port",
                "sample": f"commit+open of the first 2^{m.bit_length() - 1} coefficients of the same row "
                          f"(oracle/kzg_cpu.c); seconds by thread count: "
                          + ", ".join(f"{th}: {s:.3f}" for th, s in sorted(per.items())),
                "ms_scaled_to_full_row": per[best] * (T / m) * 1e3,
                "matches_gpu_bit_exact": cpu_res == tuple(gpu_same)}
            assert cpu_res == tuple(gpu_same), "GPU commit+open differs from the CPU oracle on the baseline sample"
        rows[f"2^{lg}"] = rec
        eng.close()
    return rows


def dist_extra_workloads(args, ctl, eng, gather, lagrange_factor, rank, world):
    """With a process group and no --workload: the OTHER multi-GPU configurations of BASELINE.json, measured in the same
    launch so that a driver SCALE run (which passes no flags) yields every multi-GPU number:
      pianist_kzg22  configs[4]: one degree-2^22 commit+open per rank (Pianist worker row `rank`), no exchange (weak)
      msm26          configs[3]: ONE 2^26-point MSM, SRS split into `world` contiguous segments, one per rank, partials
                     all_gathered and summed on every rank (strong scaling: total work fixed)
    Same timing discipline as the headline: W warm-up steps, K steps between barrier + synchronize, MAX over ranks.
    The same engine (and its communicator) serves all of them: the tables are rebuilt per workload.  Every phase runs in
    try/except on every rank and its status is agreed through the store BEFORE anybody enters a collective of the next
    phase: a failure becomes {"error": ...} in that object, never a hang and never a lost headline.  pianist first: it
    needs no data-path collective and the least memory; msm26 (largest tables, a collective per step) last."""
    res = {}
    steps, warm = max(1, min(args.steps, 10)), max(1, min(args.warmup, 3))

    def phase(name, setup, timed, report):
        """setup() -> state on every rank; agreed; timed(state) -> per-rank result inside barriers; agreed; report()."""
        if ctl.poisoned:
            res[name] = {"error": f"skipped: the process group is unusable after {ctl.poisoned}"}
            return
        state, err = None, ""
        try:
            inject(name + "_setup", rank)
            state = setup()
        except BaseException as e:       # noqa: BLE001 -- OOM, HIP errors, injected faults: all become a status
            err = errmsg(e)
        ok, bad = ctl.agree(name + "_setup", not err, err)
        if not ok:
            res[name] = {"error": "setup failed", "ranks": {str(r): m for r, m in sorted(bad.items())}}
            return
        out, err = None, ""
        try:
            out = timed(state)
        except BaseException as e:       # noqa: BLE001
            err = errmsg(e)
        ok, bad = ctl.agree(name + "_timed", not err, err)
        if not ok:
            # somebody left the timed loop early: collectives of the group may be half-done on the others
            ctl.poisoned = f"{name} failed inside its timed region"
            res[name] = {"error": "timed region failed", "ranks": {str(r): m for r, m in sorted(bad.items())}}
            return
        try:
            res[name] = report(state, out)
        except BaseException as e:       # noqa: BLE001
            res[name] = {"error": "report failed: " + errmsg(e)}
            ctl.poisoned = f"{name} failed while gathering its results"

    # ---- pianist_kzg22: worker row `rank` on this GPU, full commit+open, no data-path collective
    lg_row = args.kzg22_log
    T = 1 << lg_row
    ms = max(0, (world - 1).bit_length())
    alpha = uniform_fr(1, seed=1)

    def pianist_setup():
        t0 = time.time()
        row = uniform_fr(T, seed=rank)
        tau_y = (TAU * 7 + 1) % R_MOD
        eng.gen_srs(TAU, 0, lg_row + ms, ms, factors=[lagrange_factor(rank, ms, tau_y)])
        eng.upload_fr(0, row, True)
        ref = None
        for _ in range(warm):
            ref = eng.commit_open_resident(0, 0, T, alpha, True)
        return {"ref": ref, "setup_s": time.time() - t0, "window": eng.window}

    def pianist_timed(st):
        ctl.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            assert eng.commit_open_resident(0, 0, T, alpha, True) == st["ref"], "non-deterministic commit+open"
        ctl.barrier()
        return ctl.max_over_ranks(time.perf_counter() - t0)

    def pianist_report(st, el):
        rows = ctl.gather_bytes(b"".join(st["ref"]))          # 48 + 32 + 48 bytes per rank
        agg = eng.g1_sum_compressed(b"".join(r[:48] for r in rows))     # master aggregation: sum_i commit_i
        alg = 384.0 * T
        return {
            "metric": f"KZG commit+open coefficients/sec at 2^{lg_row} per segment", "value": T * world * steps / el,
            "unit": "coefficients/s", "ms_per_step": el / steps * 1e3, "per_segment_latency_ms": el / steps * 1e3,
            "steps": steps, "warmup": warm, "scaling": "weak", "n_gpus": world, "window_bits": st["window"],
            "workload": f"Pianist segments: {world} worker row(s) of 2^{lg_row} evaluation-form coefficients, one per GPU, full "
                        "commit+open (INTT + 2 MSM + quotient) per segment, no exchange on the data path",
            "results_hex_by_rank": [r.hex() for r in rows], "aggregate_commitment_hex": agg.hex(),
            "setup_s": round(st["setup_s"], 2),
            "roofline": {"bound": "hbm", "achieved": alg / (el / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (el / steps) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg,
                         "per": "whole commit+open call", "traffic": None}}

    phase("pianist_kzg22", pianist_setup, pianist_timed, pianist_report)

    # ---- msm26
    lg_total = args.msm26_log
    if world & (world - 1) or (1 << lg_total) < world:
        res["msm26"] = {"error": f"needs a power-of-two number of ranks <= 2^{lg_total} (got {world})"}
        return res
    n_total = 1 << lg_total
    n = n_total // world
    lg = n.bit_length() - 1

    def msm26_setup():
        if gather is None:
            raise RuntimeError("no usable collective on this rank (its engine is stuck in the communicator's init)")
        t0 = time.time()
        scal = uniform_fr(n, seed=1000 + rank)
        eng.gen_srs(TAU, 1, lg, 0, factors=[pow(TAU, rank * n, R_MOD)])
        eng.upload_fr(0, scal, False)
        return {"setup_s": time.time() - t0, "window": eng.window}

    def msm26_timed(st):
        ref = None
        for _ in range(warm):
            ref = gather.msm(0, n, 0)      # partial -> all_gather -> sum, on the device
        eng.set_profiling(2)
        try:
            ctl.barrier()
            t0 = time.perf_counter()
            acc = 0.0
            for k in range(steps):
                if k == steps // 2:
                    inject("msm26_step", rank)
                r = gather.msm(0, n, 0)
                acc += eng.timings().get("accumulate", 0.0)
                assert r == ref, "non-deterministic sharded MSM"
            ctl.barrier()
            el = ctl.max_over_ranks(time.perf_counter() - t0)
        finally:
            eng.set_profiling(0)
        return {"ref": ref, "el": el, "kernel_ms": acc / steps}

    def msm26_report(st, out):
        allr = ctl.gather_bytes(out["ref"])
        equal = all(x == allr[0] for x in allr)
        kernel_ms, el = out["kernel_ms"], out["el"]
        ach = 128.0 * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None
        rec = {
            "metric": f"BLS12-381 G1 MSM points/sec at 2^{lg_total} (SRS-sharded)", "value": n_total * steps / el,
            "unit": "points/s", "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warm, "scaling": "strong",
            "n_gpus": world, "points_per_gpu": n, "window_bits": st["window"], "result_hex": out["ref"].hex(),
            "all_ranks_equal": equal, "setup_s": round(st["setup_s"], 2),
            "workload": f"2^{lg_total}-point G1 MSM, SRS split into {world} contiguous segment(s) of 2^{lg} points, "
                        "partials all_gathered (192 B per rank), summed on every rank",
            "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS if ach else None, "kernel_ms": kernel_ms,
                         "algorithmic_bytes": 128.0 * n, "traffic": None}}
        if not equal:
            rec["error"] = "ranks disagree on the sharded MSM result"
        return rec

    phase("msm26", msm26_setup, msm26_timed, msm26_report)
    return res


def e2e_from_text_report(device, logs, with_cpu):
    """The route the reference ACTUALLY runs, at the reference's sizes (N = 1, after the timed regions, never part of
    `value`): for T = 2^16 (mainnet, Makefile:63-74), 2^12 (testnet, :89-101) and 2^10 (default flags, utils/config.py:
    152-164) -- `Client.worker_commit` then `worker_open` from List[str] (the unchanged neurons/miner.py:56-61; the second
    call is a verified row-cache hit) and the one-call `worker_commit_and_open`; median of >= 20 requests, every answer
    asserted equal to the C oracle's on the same row."""
    from zkp_subnet_amd import codec
    from zkp_subnet_amd.client import Client

    rows = {}
    for lg in logs:
        T = 1 << lg
        cl = Client(seed=3, workers=[0], device=device)
        cl.start(scale=lg, machines_scale=0)
        try:
            # SIX different rows in rotation: the library keeps the last four rows' coefficients, so every worker_commit is
            # a miss (as for a fresh challenge) and every worker_open that follows it a verified hit -- what a miner sees
            nrows = 6
            raws = [uniform_fr(T, seed=100 * lg + k) for k in range(nrows)]
            polys = [codec.be32_to_fr_list(r) for r in raws]
            xb = uniform_fr(1, seed=2)
            x = codec.be32_to_fr(xb)
            wants = None
            if with_cpu:
                from oracle import cpu as oc     # the checker: never inside a timed call

                oc.build()
                srs = cl.engine.srs_read(0, T)
                wants = []
                for r in raws:
                    ev, pf = oc.open_(srs, r, xb, True, threads=8)
                    wants.append({"commitment": codec.g1_to_b64(oc.commit(srs, r, True, threads=8)),
                                  "eval": codec.be32_to_fr(ev), "proof": codec.g1_to_b64(pf)})

            def two_call(poly):
                with cl.worker_commit(0, poly) as a, cl.worker_open(0, poly, x) as b:
                    assert a.status_code == 200 and b.status_code == 200, (a.json(), b.json())
                    return {"commitment": a.json()["commitment"], "eval": b.json()["eval"], "proof": b.json()["proof"]}

            def fused(poly):
                with cl.worker_commit_and_open(0, poly, x) as r:
                    assert r.status_code == 200, r.json()
                    return dict(r.json())

            rec = {"log2_T": lg}
            for name, fn in (("two_call", two_call), ("fused", fused)):
                t_w, warm = time.perf_counter(), 0
                while warm < nrows or time.perf_counter() - t_w < 0.08:      # the clocks need ~40 ms of load
                    fn(polys[warm % nrows])
                    warm += 1
                h0, m0 = cl.engine.row_cache_stats()
                lat = []
                for k in range(warm, warm + 30):     # the rotation goes on where the warm-up left it: no row is still cached
                    t1 = time.perf_counter()
                    got = fn(polys[k % nrows])
                    lat.append((time.perf_counter() - t1) * 1e3)
                    if wants is not None:
                        assert got == wants[k % nrows], f"{name} route at 2^{lg} differs from the C oracle"
                h1, m1 = cl.engine.row_cache_stats()
                rec[name + "_ms"] = {"median": round(pctl(lat, 0.5), 4), "p10": round(pctl(lat, 0.1), 4),
                                     "p90": round(pctl(lat, 0.9), 4), "requests": len(lat)}
                if name == "two_call":
                    rec["two_call_row_cache_hits_misses"] = [h1 - h0, m1 - m0]   # every worker_open a verified hit
            rec["matches_cpu_oracle_bit_exact"] = wants is not None
            rec["wire_codec"] = "csrc/wire_py.c (AVX2, pinned staging)" if codec._wire is not None else "python"
            rows[f"2^{lg}"] = rec
        finally:
            cl.stop()
    return rows
