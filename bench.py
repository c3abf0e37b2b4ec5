#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the KZG segment-prover hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload msm20|kzg22]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json configs[1], the one the metric is quoted on): one step = one 2^20-point BLS12-381 G1
Pippenger MSM per GPU over a cached (device-resident) SRS segment with device-resident random scalars.  With N > 1
the MSM is SRS-sharded: rank g owns segment [g*2^20, (g+1)*2^20) of a 2^20*N-point SRS, reduces it to one 192-byte
partial on its GPU, the partials are all_gathered over RCCL/xGMI and summed on every rank (weak scaling: per-GPU work
fixed).  `value` = points of all ranks / wall time of the K timed steps (max over ranks).

Extra objects on the JSON line: `roofline` for the dominant kernel (k_msm_accumulate, duration from HIP events on the
library's own stream, denominators from SURVEY.md 8d: 128 B per point) and `cpu_baseline` (oracle/kzg_cpu.c, a C port
timed on this box's host cores on a bounded sample of the same inputs; also used here as a parity check).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy peak
HBM_COPY_GBS = 6290.0
VALU_PEAK_GINST_S = 256 * 4 * 2.4 / 4   # wave64 VALU instructions per ns, chip-wide: one per 4 cycles per SIMD at 2.4 GHz
TAU = 0x2F6C7A1D3B5E9F80412D6A7C93E1B5F7086A4D2C1E9B3F5A7D6C8E0F1A2B3C4D % R_MOD


def uniform_fr(n, seed):
    """n scalars uniform in [0, r): seeded PCG64 stream, 255-bit candidates, rejection of values >= r."""
    import numpy as np

    rng = np.random.default_rng(seed)
    r_words = np.array([(R_MOD >> (64 * (3 - i))) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    out = []
    have = 0
    while have < n:
        m = int((n - have) * 1.15) + 64
        raw = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
        raw[:, 0] &= 0x7F
        w = raw.view(">u8").astype(np.uint64)
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for i in range(4):
            lt |= eq & (w[:, i] < r_words[i])
            eq &= w[:, i] == r_words[i]
        keep = raw[lt]
        out.append(keep)
        have += len(keep)
    return np.concatenate(out)[:n].tobytes()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measured_copy_peak_gbs(torch):
    """STREAM-style device copy (1 GiB read + 1 GiB written per pass) on torch's stream: the achievable-HBM yardstick
    SURVEY 8d asks for beside the 8 TB/s spec figure."""
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return gbs


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="msm20", choices=["msm20", "kzg22"])
    ap.add_argument("--log-n", type=int, default=0, help="override log2(points per GPU) (debug)")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--cpu-sample-log", type=int, default=19)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(host cores, 16): the box's CPU share for one GPU")
    ap.add_argument("--in-flight", type=int, default=1, choices=[1, 2],
                    help="msm20: MSM requests in flight per GPU in the timed region (1 = one request at a time, the "
                         "default; 2 = the library's two lanes)")
    ap.add_argument("--pipelined", action="store_true",
                    help="msm20: after the timed region, time the same K steps again with two requests in flight and "
                         "report them as `pipelined` (off by default so that every k_msm_accumulate launch of the "
                         "default run is an uncontended one and rocprofv3's average matches `roofline.kernel_ms`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adversarial", action="store_true", help="skip the separately reported adversarial inputs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL group (self-test)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from zkp_subnet_amd import HipEngine
    from zkp_subnet_amd.distributed import all_gather_partials

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    eng = HipEngine(local_rank, window=args.window)
    t_setup = time.time()
    if args.workload == "msm20":
        lg = args.log_n or 20
        n = 1 << lg
        # this rank's SRS segment: points [rank*n, (rank+1)*n) of the 2^lg * world point SRS [tau^j] G
        eng.gen_srs(TAU, 1, lg, 0, factors=[pow(TAU, rank * n, R_MOD)])
        scal = uniform_fr(n, seed=rank)                       # seed 0 on rank 0
        eng.upload_fr(0, scal, False)
        alpha = None
    else:
        lg = args.log_n or 22
        n = 1 << lg
        # Pianist segments: worker row `rank`, one per GPU, no exchange (BASELINE.json configs[2] / [4])
        from zkp_subnet_amd.engine import lagrange_factor
        ms = max(0, (world - 1).bit_length())
        eng.gen_srs(TAU, 0, lg + ms, ms, factors=[lagrange_factor(rank, ms, (TAU * 7 + 1) % R_MOD)])
        scal = uniform_fr(n, seed=rank)
        eng.upload_fr(0, scal, True)
        alpha = uniform_fr(1, seed=1)
    setup_s = time.time() - t_setup
    plan = eng.msm_plan(n)

    results = []
    stage_sum = {}
    step_ms = []
    depth = args.in_flight if args.workload == "msm20" else 1
    state = {"depth": depth}

    # N > 1, one request at a time: partial -> RCCL all_gather -> sum entirely through device buffers
    gather = None
    if use_dist and args.workload == "msm20" and depth == 1:
        from zkp_subnet_amd.distributed import DeviceGather

        gather = DeviceGather(eng)
    state["gather"] = gather

    def submit():
        if args.workload == "msm20" and state["gather"] is None:
            return eng.msm_submit(0, n, 0, partial=use_dist)
        return None

    def complete(ticket, collect):
        if state["gather"] is not None:
            r = state["gather"].msm(0, n, 0)
            if collect:
                for k, v in eng.timings().items():
                    stage_sum[k] = stage_sum.get(k, 0.0) + v
        elif args.workload == "msm20":
            r = eng.msm_wait(ticket)
            if collect:
                for k, v in eng.timings().items():
                    stage_sum[k] = stage_sum.get(k, 0.0) + v
            if use_dist:
                r = eng.g1_sum(b"".join(all_gather_partials(r)))
        else:
            r = eng.commit_open_resident(0, 0, n, alpha, True)
            if collect:
                for k, v in eng.timings().items():
                    stage_sum[k] = stage_sum.get(k, 0.0) + v
        results.append(r)

    def run_steps(count, collect):
        """`count` steps with up to `depth` requests in flight; every step's result reaches the host inside the loop."""
        pending = []
        last = time.perf_counter()
        for _ in range(count):
            pending.append(submit())
            if len(pending) == state["depth"]:
                complete(pending.pop(0), collect)
                now = time.perf_counter()
                step_ms.append((now - last) * 1e3)
                last = now
        while pending:
            complete(pending.pop(0), collect)
            now = time.perf_counter()
            step_ms.append((now - last) * 1e3)
            last = now

    run_steps(args.warmup, False)
    results.clear()
    step_ms.clear()
    # msm20 and short rows (one batched pass): stage spans (HIP events on the library's stream) are recorded inside the
    # timed region.  Rows above 2^18: the library runs the two MSMs of a commit+open on two streams unless profiling is
    # on, so the timed region runs unprofiled and the stage times come from extra profiled (serialised) steps afterwards.
    profile_in_timed = args.workload == "msm20" or n <= (1 << 18)
    eng.set_profiling(profile_in_timed)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, profile_in_timed)
    barrier()
    elapsed = time.perf_counter() - t0
    n_prof = args.steps
    timed_step_ms = list(step_ms)
    if not profile_in_timed:
        eng.set_profiling(True)
        n_prof = min(args.steps, 5)
        run_steps(n_prof, True)
    eng.set_profiling(False)
    # the same K steps with two requests in flight on the library's two lanes (MSM i+1's sort/accumulate overlaps the
    # latency-bound tail of MSM i): reported beside the headline, never mixed into it
    pipelined = None
    if args.workload == "msm20" and depth == 1 and args.pipelined:
        state["depth"], state["gather"] = 2, None
        run_steps(args.warmup, False)
        barrier()
        tp = time.perf_counter()
        run_steps(args.steps, False)
        barrier()
        pipe_s = time.perf_counter() - tp
        state["depth"], state["gather"] = 1, gather
        if use_dist:
            t = torch.tensor([pipe_s], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pipe_s = float(t.item())
        pipelined = {"requests_in_flight": 2, "value": n * world * args.steps / pipe_s, "unit": "points/s",
                     "ms_per_step": pipe_s / args.steps * 1e3}
    # single-request latency (one request at a time, result on the host before the next starts)
    lat = []
    for _ in range(min(args.steps, 10)):
        tl = time.perf_counter()
        if args.workload == "msm20":
            if not use_dist:
                results.append(eng.msm_resident(0, n, 0))
            elif gather is not None:
                results.append(gather.msm(0, n, 0))
            else:
                results.append(eng.g1_sum(b"".join(all_gather_partials(eng.msm_partial_resident(0, n, 0)))))
        else:
            results.append(eng.commit_open_resident(0, 0, n, alpha, True))
        lat.append((time.perf_counter() - tl) * 1e3)
    latency_ms = sorted(lat)[len(lat) // 2]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(r == results[0] for r in results), "non-deterministic result across steps"

    if rank == 0:
        stages = {k: v / n_prof for k, v in stage_sum.items()}
        sm = sorted(timed_step_ms)
        pct = lambda q: sm[min(len(sm) - 1, int(q * len(sm)))]
        units = n * world * args.steps
        acc_ms = stages.get("accumulate", 0.0)
        if args.workload == "msm20":
            alg_bytes = 128.0 * n                 # 96 B affine point + 32 B scalar, each read once
            launches = 1
            metric, unit, value = "BLS12-381 G1 MSM points/sec at 2^20", "points/s", units / elapsed
            wl = f"2^{lg}-point BLS12-381 G1 Pippenger MSM per GPU (uniform scalars in [0,r), cached SRS)"
        else:
            batched = n <= (1 << 18)              # short rows: ONE accumulate launch carries both MSMs
            alg_bytes = (2 if batched else 1) * 128.0 * n   # per k_msm_accumulate launch (commit: n, open: n-1 scalars)
            launches = 1 if batched else 2
            metric, unit, value = "KZG commit+open coefficients/sec at 2^22", "coefficients/s", units / elapsed
            wl = f"degree-2^{lg} KZG commit+open per GPU (INTT + 2 MSM + quotient), evaluation-form input"
        copy_gbs = measured_copy_peak_gbs(torch) if world == 1 else None
        per_launch_s = acc_ms / 1e3 / launches if acc_ms else float("nan")
        achieved = alg_bytes / per_launch_s / 1e9 if acc_ms else None
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the value is
        # the one measured with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 FETCH correction)
        # on this exact configuration and committed under profiles/; null for any other configuration.
        traffic = None
        valu_insts = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            if (pmc["workload"], pmc["points_per_gpu"], pmc["window_bits"]) == (args.workload, n, eng.window):
                traffic = pmc["traffic_bytes_per_launch"]
                valu_insts = pmc.get("sq_insts_valu_per_launch")
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl, "points_per_gpu": n, "window_bits": eng.window, "windows": plan["windows"],
                       "buckets": plan["buckets"], "entries_per_lane": plan["chunk"], "lanes": plan["lanes"],
                       "requests_in_flight": depth,
                       "parallelism": "single GPU" if world == 1 else
                       (f"SRS-sharded x{world}, all_gather of 192 B partials over RCCL" if args.workload == "msm20"
                        else f"Pianist segments x{world}, no exchange")},
            "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel_ms": per_launch_s * 1e3 if acc_ms else None, "algorithmic_bytes": alg_bytes,
                         "measured_copy_peak_gbs": round(copy_gbs, 1) if copy_gbs else None,
                         "frac_of_measured_copy_peak": (achieved / (copy_gbs or HBM_COPY_GBS)) if achieved else None,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc, 2 x FETCH_SIZE + WRITE_SIZE)" if traffic else None,
                         "note": "integer-VALU-bound (14 x 28-bit-limb Montgomery products on v_mad_u64_u32: ~3.7k mads "
                                 "per mixed point addition), not HBM-bound; see DESIGN.md 3.3"},
            # what actually bounds the kernel: VALU wave-instructions issued per second against one
            # wave64 instruction per 4 cycles per SIMD (256 CUs x 4 SIMDs x 2.4 GHz / 4); SQ_INSTS_VALU from the same
            # rocprofv3 --pmc passes as `traffic`
            "valu_issue": ({"wave_insts_per_launch": valu_insts,
                            "achieved_ginst_s": valu_insts / per_launch_s / 1e9,
                            "peak_ginst_s": VALU_PEAK_GINST_S,
                            "frac": valu_insts / per_launch_s / 1e9 / VALU_PEAK_GINST_S}
                           if valu_insts and acc_ms else None),
            "result_hex": results[0].hex() if isinstance(results[0], (bytes, bytearray)) else b"".join(results[0]).hex(),
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "single_request_latency_ms": round(latency_ms, 4),
            "pipelined": pipelined,
            "step_ms": {"median": round(pct(0.5), 4), "p10": round(pct(0.1), 4), "p90": round(pct(0.9), 4)},
            "setup_s": round(setup_s, 2),
        }
        if args.workload == "kzg22":
            out["kzg_commit_open_latency_ms"] = latency_ms
            if world == 1:      # the same call fed from host memory: + H2D of the row (never part of `value`)
                tp = time.perf_counter()
                for _ in range(3):
                    assert eng.commit_open(0, scal, alpha, True) == results[0]
                out["pcie_inclusive_latency_ms"] = round((time.perf_counter() - tp) / 3 * 1e3, 4)
        # ---- adversarial scalar distributions (SURVEY 8d cfg 2: reported separately, never part of `value`)
        if world == 1 and args.workload == "msm20" and not args.no_adversarial:
            import numpy as np

            adv = {}
            same = uniform_fr(1, seed=77) * n
            small = np.zeros((n, 32), dtype=np.uint8)
            small[:, 28:] = np.random.default_rng(78).integers(0, 256, size=(n, 4), dtype=np.uint8)
            for name, data in (("all_equal", same), ("below_2^32", small.tobytes())):
                eng.upload_fr(1, data, False)
                eng.msm_resident(1, n, 0)
                ta = time.perf_counter()
                for _ in range(5):
                    eng.msm_resident(1, n, 0)
                adv[name] = {"ms_per_msm": round((time.perf_counter() - ta) / 5 * 1e3, 4)}
            out["adversarial"] = adv
        # ---- CPU baseline: the oracle's C port on a bounded sample of the same inputs, all host cores
        if world == 1 and not args.no_cpu_baseline and args.workload == "msm20":
            from oracle import cpu as oc

            oc.build()
            m = 1 << min(args.cpu_sample_log, lg)
            cores = args.cpu_threads or min(host_cores(), 16)
            srs = eng.srs_read(0, m)
            prep = oc.PreparedMsm(srs, scal[: 32 * m])
            tc = time.perf_counter()
            cpu_res = prep.run(cores)
            cpu_s = time.perf_counter() - tc
            tc1 = time.perf_counter()
            prep1 = oc.PreparedMsm(srs[: 96 * (m >> 3)], scal[: 32 * (m >> 3)])
            prep1.run(1)
            cpu1_s = time.perf_counter() - tc1
            gpu_same = eng.msm(scal[: 32 * m], 0)
            out["cpu_baseline"] = {
                "value": m / cpu_s, "unit": "points/s", "cores": cores, "kind": "port",
                "sample": f"first 2^{m.bit_length() - 1} points/scalars of the same workload, one Pippenger MSM split over "
                          f"{cores} threads ({cpu_s:.2f} s wall, {host_cores()} host cores visible, {cpu_model()}); 1 thread on "
                          f"2^{(m >> 3).bit_length() - 1}: {(m >> 3) / cpu1_s:.0f} points/s",
                "single_thread_points_per_s": (m >> 3) / cpu1_s,
                "matches_gpu_bit_exact": cpu_res == gpu_same,
            }
            assert cpu_res == gpu_same, "GPU MSM differs from the CPU oracle on the baseline sample"
        # RCCL writes its version banner through C stdio, which is flushed at exit: push it out now so that the JSON
        # line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
