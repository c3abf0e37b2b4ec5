#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the KZG segment-prover hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload msm20|msm26|kzg22]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload `msm20` (BASELINE.json configs[1], the one the metric is quoted on): one step = one 2^20-point
BLS12-381 G1 Pippenger MSM per GPU over a cached (device-resident) SRS segment with device-resident random scalars.
With N > 1 the MSM is SRS-sharded: rank g owns segment [g*2^20, (g+1)*2^20) of a 2^20*N-point SRS, reduces it to one
192-byte partial on its GPU, the partials are all_gathered over RCCL/xGMI and summed on every rank (weak scaling:
per-GPU work fixed).  `value` = points of all ranks / wall time of the K timed steps (max over ranks).

`--workload msm26` is BASELINE.json configs[3]: ONE 2^26-point MSM whose SRS is split into N contiguous segments,
one per rank (N = 1: all 2^26 points and 103 GB of window tables on one GPU); strong scaling, same collective.
`--workload kzg22` is configs[2] / [4]: one degree-2^22 commit+open per GPU (N > 1: Pianist rows, no exchange).

With a process group (N > 1, the driver's SCALE runs) and no --workload, the SAME launch also measures configs[3] and
configs[4] after the headline region and adds them to the line as `msm26` (one 2^26-point MSM over N SRS segments,
strong scaling) and `pianist_kzg22` (one 2^22 commit+open per rank): a flag-less multi-GPU run yields every BASELINE
multi-GPU number.  Every rank's MSM result is checked against rank 0's.

Robustness of a multi-rank launch (the driver's SCALE runs cannot be watched or repeated):
  * the control plane (barriers, MAX over ranks, cross-rank result checks) is a gloo process group with a timeout; phase
    status travels through the rendezvous STORE, so one rank's failure can never park the others in a collective;
  * the data path is the LIBRARY's own collective (kzg_comm_init / kzg_msm_sharded: one ncclAllGather of 192 B per rank on
    the lane's stream, RCCL over xGMI).  Preflight before any table is built: communicator + a checked all_gather, both
    under a watchdog; if any rank fails it every rank falls back to the all_gather of the gloo group (device tensors
    through the host) and the line says so in `config.collective` / `config.rccl_version` -- a curve exists either way,
    the exchange is 192 B per rank;
  * rank 0 prints the headline line as soon as the headline region and its CPU baseline are done, and the augmented line
    (msm26, pianist_kzg22) after the extras: the LAST line is the record, the first is the insurance;
  * every extra runs in try/except on every rank, failures become {"msm26": {"error": ...}} and the exit code stays 0;
  * `python bench.py --gpus N` (no launcher) is watched by its parent: BENCH_WATCHDOG_S (default 1500) seconds, then the
    exact child it started is terminated and the last line seen is still printed.
BENCH_BACKEND=nccl + BENCH_COLLECTIVE=torch is the round-4 form (torch's RCCL group carries everything): the A/B.
BENCH_FAULT=<phase>:<rank> (msm26_setup, msm26_step, pianist_kzg22_setup, comm_init) injects a failure for the tests.

Extra objects on the JSON line:
  roofline         dominant kernel (k_msm_accumulate): duration from HIP events on the library's own stream inside the
                   timed region, denominators from SURVEY.md 8d (128 B per point)
  mad_issue        what actually bounds that kernel: v_mad_u64_u32 wave-instructions per second against the chip's
                   measured mad rate (profiles/ubench_valu_rates.txt) -- see DESIGN.md 3.3
  e2e_from_text    N = 1 default run only: the route the reference actually runs (neurons/miner.py:56-61: worker_commit then
                   worker_open, List[str] in) and the fused call, at mainnet 2^16, testnet 2^12 and default-flag 2^10 rows,
                   median of >= 20, each answer asserted equal to the C oracle's on the same row
  kzg_commit_open  N = 1 default run only: commit+open latency of device-resident rows of 2^22 (configs[2]), 2^16
                   (mainnet row) and 2^12 (testnet row) coefficients, each with p10/p90, stage times, a roofline entry at
                   384 B per coefficient and its own CPU baseline
  cpu_baseline     oracle/kzg_cpu.c (a C port; the real prover is an absent Rust binary) timed on this box's host
                   cores on a bounded sample of the same inputs: 1 thread, 16 threads (the box's share per GPU) and all
                   visible cores; also a parity check of the GPU result
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import *  # noqa: E402,F401,F403  (scripts import TAU / uniform_fr / source_sha16 from here)
from benchlib.common import (HBM_COPY_GBS, HBM_PEAK_GBS, MAD_NS, R_MOD, SIMDS, TAU, VALU_FULL_RATE_GINST_S,  # noqa: E402,F401
                             cpu_model, errmsg, flush_c_stdio, host_cores, identity, measured_copy_peak_gbs, pctl,
                             source_sha16, sysfs_sclk_mhz, thread_counts, uniform_fr)
from benchlib.control import Ctl, Fault, inject, make_collective, self_launch  # noqa: E402,F401
from benchlib.extras import dist_extra_workloads, e2e_from_text_report, kzg_rows_report  # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="msm20", choices=["msm20", "msm26", "kzg22"])
    ap.add_argument("--log-n", type=int, default=0,
                    help="override log2(points per GPU) (msm20, kzg22) or log2(total points) (msm26)")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--cpu-sample-log", type=int, default=20)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = 1, 16 and all visible cores")
    ap.add_argument("--in-flight", type=int, default=1, choices=[1, 2],
                    help="MSM requests in flight per GPU in the timed region (1 = one request at a time, the default; "
                         "2 = two of the library's lanes)")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the second timing of the same K steps with two requests in flight (`pipelined`)")
    ap.add_argument("--no-kzg-rows", action="store_true", help="skip the commit+open latency rows (`kzg_commit_open`)")
    ap.add_argument("--kzg-rows", default="22,16,12,10", help="log2 row lengths of `kzg_commit_open`")
    ap.add_argument("--no-e2e", action="store_true", help="skip the from-text rows of the reference's own route (`e2e_from_text`)")
    ap.add_argument("--e2e-rows", default="16,12,10", help="log2 row lengths of `e2e_from_text`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adversarial", action="store_true", help="skip the separately reported adversarial inputs")
    ap.add_argument("--no-dist-extra", action="store_true",
                    help="with a process group and the default workload: skip the msm26 / pianist_kzg22 objects")
    ap.add_argument("--msm26-log", type=int, default=26, help="log2(total points) of the `msm26` object")
    ap.add_argument("--kzg22-log", type=int, default=22, help="log2(row length) of the `pianist_kzg22` object")
    ap.add_argument("--headline-only", action="store_true",
                    help="timed region only (profiling runs: every k_msm_accumulate launch is then an uncontended launch "
                         "of the headline size, so rocprofv3's average matches `roofline.kernel_ms`)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_pipelined = args.no_kzg_rows = args.no_cpu_baseline = args.no_adversarial = args.no_dist_extra = args.no_e2e = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` with no launcher: start the driver's own launch line as a CHILD and relay it.  Nothing
        # in this parent has touched torch or HIP (a process that has initialised the GPU must never be replaced or forked
        # into ranks); the child is fresh, its stdout is relayed line by line so the JSON line stays the last one.
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    # eight hardware queues (the library's default, zkp_subnet_amd/_native.py): torch.cuda below is this process's first HIP
    # user, so the variable has to be in place before it -- the library loads later
    # (the one-GPU self-test puts several RANKS on one device: processes that share a GPU keep the runtime's four -- with eight
    # queues each, two busy processes oversubscribe the hardware queues and every cross-stream wait costs a scheduling quantum:
    # 33.5 against 6.4 ms per step, profiles/r05_ab_hw_queues.log)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if os.environ.get("BENCH_ONE_GPU") == "1" else "8")
    if world > 1:
        # ONE node by contract (the launch line is --nnodes=1, rendezvous on 127.0.0.1): the bootstrap sockets of RCCL and
        # gloo go over loopback whatever other interfaces the container shows and whether or not its hostname resolves
        # (data still moves over xGMI / P2P).  Respected if the caller has set them.
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # The process group is the CONTROL plane (barriers, MAX over ranks, cross-rank checks, the rendezvous of the library's
    # communicator): gloo with a timeout by default.  The data path of the sharded MSM is the library's own ncclAllGather
    # (RCCL over xGMI), see make_collective().  BENCH_BACKEND=nccl: torch's RCCL group carries both (the round-4 form).
    # BENCH_ONE_GPU=1: every rank on device 0 -- the self-test of the N > 1 logic (rank-dependent SRS segments, cross-rank
    # checks, msm26 / pianist_kzg22, the fallback from a communicator that cannot form) on a one-GPU box, where RCCL refuses
    # two ranks on one device (tests/test_gpu_bench.py).  Never what a measurement uses.
    backend = os.environ.get("BENCH_BACKEND", "gloo")
    if os.environ.get("BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the latter: a 1-rank group (self-test)
    pg_timeout_s = float(os.environ.get("BENCH_PG_TIMEOUT_S", "180"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=pg_timeout_s), **kw)
    ctl = Ctl(torch, dist, rank, world, use_dist, pg_timeout_s + 90)
    barrier = ctl.barrier

    from zkp_subnet_amd import HipEngine
    from zkp_subnet_amd.distributed import all_gather_partials
    from zkp_subnet_amd.engine import lagrange_factor

    is_msm = args.workload in ("msm20", "msm26")
    eng = HipEngine(local_rank, window=args.window)
    # the collective of the sharded MSM is decided NOW, before any table is built or scalar uploaded: communicator +
    # checked all_gather under a watchdog, agreed between the ranks through the store; a failure falls back, flagged
    gather, coll_info, leaked = None, None, []
    if use_dist and is_msm:
        gather, coll_info = make_collective(args, ctl, eng, torch, dist, rank, world)
        if coll_info.get("comm_init_helper_left_behind"):
            leaked.append("a join thread inside RCCL's rendezvous")
    t_setup = time.time()
    if args.workload == "msm20":
        lg = args.log_n or 20
        n = 1 << lg
        n_total = n * world
        # this rank's SRS segment: points [rank*n, (rank+1)*n) of the 2^lg * world point SRS [tau^j] G
        scal = uniform_fr(n, seed=rank)                       # seed 0 on rank 0; host work first: the GPU
        eng.gen_srs(TAU, 1, lg, 0, factors=[pow(TAU, rank * n, R_MOD)])   # goes from the table build straight into
        eng.upload_fr(0, scal, False)                         # the warm-up steps (an idle gap drops its clocks)
        alpha = None
        scaling = "weak"
    elif args.workload == "msm26":
        lg_total = args.log_n or 26
        if world & (world - 1) or (1 << lg_total) % world:
            raise SystemExit("msm26 needs a power-of-two number of ranks")
        n_total = 1 << lg_total
        n = n_total // world
        lg = n.bit_length() - 1
        # contiguous SRS segment [rank*n, (rank+1)*n) of the 2^26-point SRS; scalars of the same index range
        scal = uniform_fr(n, seed=1000 + rank)
        eng.gen_srs(TAU, 1, lg, 0, factors=[pow(TAU, rank * n, R_MOD)])
        eng.upload_fr(0, scal, False)
        alpha = None
        scaling = "strong"
    else:
        lg = args.log_n or 22
        n = 1 << lg
        n_total = n * world
        # Pianist segments: worker row `rank`, one per GPU, no exchange (BASELINE.json configs[2] / [4])
        ms = max(0, (world - 1).bit_length())
        scal = uniform_fr(n, seed=rank)
        alpha = uniform_fr(1, seed=1)
        eng.gen_srs(TAU, 0, lg + ms, ms, factors=[lagrange_factor(rank, ms, (TAU * 7 + 1) % R_MOD)])
        eng.upload_fr(0, scal, True)
        scaling = "weak"
    setup_s = time.time() - t_setup
    plan = eng.msm_plan(n)

    results = []
    stage_sum = {}
    step_ms = []
    depth = args.in_flight if is_msm else 1
    state = {"depth": depth}

    # N > 1, one request at a time: partial -> all_gather -> sum entirely on the device (make_collective above)
    if depth != 1:
        gather_step = None
    else:
        gather_step = gather
    state["gather"] = gather_step

    def submit():
        if is_msm and state["gather"] is None:
            return eng.msm_submit(0, n, 0, partial=use_dist)
        return None

    def complete(ticket, collect):
        if state["gather"] is not None:
            r = state["gather"].msm(0, n, 0)
            if collect:
                for k, v in eng.timings().items():
                    stage_sum[k] = stage_sum.get(k, 0.0) + v
        elif is_msm:
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

    # the same K steps with two requests in flight on two of the library's lanes (MSM i+1's sort/accumulate overlaps the
    # latency-bound tail of MSM i): reported beside the headline, never mixed into it.  Measured FIRST: the GPU takes
    # ~40 ms of sustained load to reach its clocks (scripts/clock_ramp.py: steps 0-15 after an idle gap run 60-2 % slow),
    # more than W = 5 warm-up steps give it; this way the one-at-a-time region below starts on a warm GPU
    pipelined = None
    if is_msm and depth == 1 and not args.no_pipelined:
        state["depth"], state["gather"] = 2, None
        # >= 32 steps of the same load first (>= 80 ms at 2^20): the clocks need ~40 ms, and this region used to be the cold
        # one of the run.  A COUNT, never a duration: with N > 1 every step holds a collective, so all
        # ranks must run the same number of them.  Reported as `msms_before_region`.
        pipe_warm = max(args.warmup, 32)
        run_steps(pipe_warm, False)
        barrier()
        tp = time.perf_counter()
        run_steps(args.steps, False)
        barrier()
        pipe_s = time.perf_counter() - tp
        state["depth"], state["gather"] = 1, gather_step
        pipe_s = ctl.max_over_ranks(pipe_s)
        pipelined = {"requests_in_flight": 2, "value": n_total * args.steps / pipe_s, "unit": "points/s",
                     "ms_per_step": pipe_s / args.steps * 1e3, "msms_before_region": pipe_warm}
    msms_before = len(results)             # MSMs this GPU ran before the declared warm-up (the `pipelined` measurement)
    run_steps(args.warmup, False)
    results.clear()
    step_ms.clear()
    # The timed region carries HIP events around the dominant kernel only (profiling level 2: two events per
    # k_msm_accumulate launch, on the stream it is launched on -- the roofline's kernel time is measured live over exactly
    # these K steps).  A full set of stage spans costs ~0.09 ms per 2^20 MSM (every event is a barrier packet on the
    # stream) and pins concurrent calls to one lane, so the per-stage table comes from extra, serialised steps afterwards.
    live_level = 2 if is_msm and depth == 1 else 0
    eng.set_profiling(live_level)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, live_level == 2)
    barrier()
    elapsed = time.perf_counter() - t0
    # the bound of the dominant kernel, measured NOW on this box at the clocks the timed region left behind (~2 ms: a
    # v_mad_u64_u32 chain at 2 waves per SIMD on every SIMD, csrc/calibrate.hip): mad_issue.frac is a same-run ratio
    calib = None
    if is_msm:
        try:
            # the clock the SIMDs ran at is calib["memtime_ticks_per_ns"] (s_memtime over the kernel itself); the sysfs
            # pp_dpm_sclk node lags by hundreds of milliseconds (it showed 158 MHz, 1941 MHz and 2398 MHz around identical
            # kernels) and is only logged by scripts/clock_state.py
            calib = eng.calibrate(2)
        except Exception as e:                 # noqa: BLE001 -- a measurement aid must never cost the bench line
            calib = {"error": f"{type(e).__name__}: {e}"}
    acc_live_ms = stage_sum.get("accumulate", 0.0) / args.steps if live_level == 2 else None
    stage_sum.clear()
    timed_step_ms = list(step_ms)
    state["depth"] = 1
    eng.set_profiling(1)
    n_prof = min(args.steps, 5)
    run_steps(n_prof, True)
    state["depth"] = depth
    eng.set_profiling(0)
    # single-request latency (one request at a time, result on the host before the next starts)
    lat = []
    for _ in range(min(args.steps, 10)):
        tl = time.perf_counter()
        if is_msm:
            if not use_dist:
                results.append(eng.msm_resident(0, n, 0))
            elif gather_step is not None:
                results.append(gather_step.msm(0, n, 0))
            else:
                results.append(eng.g1_sum(b"".join(all_gather_partials(eng.msm_partial_resident(0, n, 0)))))
        else:
            results.append(eng.commit_open_resident(0, 0, n, alpha, True))
        lat.append((time.perf_counter() - tl) * 1e3)
    latency_ms = sorted(lat)[len(lat) // 2]
    elapsed = ctl.max_over_ranks(elapsed)
    assert all(r == results[0] for r in results), "non-deterministic result across steps"
    window_bits = eng.window
    dist_extra = None
    rccl_version = coll_info.get("rccl_version") if coll_info else None
    if use_dist and is_msm:                    # every rank computed the same sum: check it, do not assume it
        allr = ctl.gather_bytes(results[0])
        assert all(x == results[0] for x in allr), "ranks disagree on the MSM result"
    # inputs of the CPU baseline (rank 0, every N): a bounded sample of THIS rank's segment and the GPU's answer on it,
    # taken before the engine is closed; the CPU itself is timed after every timed region of the launch
    cpu_in = None
    if rank == 0 and not args.no_cpu_baseline:
        if is_msm:
            m = 1 << min(args.cpu_sample_log, lg)
            cpu_in = (m, eng.srs_read(0, m), eng.msm(scal[: 32 * m], 0))
        else:
            m = min(n, 1 << 17)
            cpu_in = (m, eng.srs_read(0, m), results[0] if m == n else eng.commit_open(0, scal[: 32 * m], alpha, True))
    out = None
    if rank == 0:
        stages = {k: v / n_prof for k, v in stage_sum.items()}
        acc_ms = acc_live_ms if acc_live_ms else stages.get("accumulate", 0.0)   # live over the timed region when available
        if is_msm:
            units = n_total * args.steps
            alg_bytes = 128.0 * n                 # 96 B affine point + 32 B scalar, each read once
            launches = 1
            metric, unit, value = "BLS12-381 G1 MSM points/sec at 2^20", "points/s", units / elapsed
            if args.workload == "msm20":
                wl = f"2^{lg}-point BLS12-381 G1 Pippenger MSM per GPU (uniform scalars in [0,r), cached SRS)"
            else:
                metric = "BLS12-381 G1 MSM points/sec at 2^26 (SRS-sharded)"
                wl = (f"2^{n_total.bit_length() - 1}-point BLS12-381 G1 Pippenger MSM, SRS split into {world} contiguous "
                      f"segment(s) of 2^{lg} points, one per GPU (uniform scalars in [0,r), cached SRS)")
        else:
            units = n_total * args.steps
            batched = n <= (1 << 18)              # short rows: ONE accumulate launch carries both MSMs
            alg_bytes = (2 if batched else 1) * 128.0 * n   # per k_msm_accumulate launch (commit: n, open: n-1 scalars)
            launches = 1 if batched else 2
            metric, unit, value = "KZG commit+open coefficients/sec at 2^22", "coefficients/s", units / elapsed
            wl = f"degree-2^{lg} KZG commit+open per GPU (INTT + 2 MSM + quotient), evaluation-form input"
        copy_gbs = measured_copy_peak_gbs(torch) if world == 1 else None
        per_launch_s = acc_ms / 1e3 / launches if acc_ms else float("nan")
        achieved = alg_bytes / per_launch_s / 1e9 if acc_ms else None
        # HBM traffic and instruction counts of the dominant kernel: PMC counters cannot be read from inside this process,
        # so the values are the ones measured with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE / SQ passes, gfx950
        # x2 FETCH correction) on this exact configuration and committed under profiles/; null for any other configuration.
        traffic = valu_insts = mads = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            if (pmc["workload"], pmc["points_per_gpu"], pmc["window_bits"]) == ("msm20" if is_msm else args.workload, n, window_bits):
                traffic = pmc["traffic_bytes_per_launch"]
                valu_insts = pmc.get("sq_insts_valu_per_launch")
                mads = pmc.get("wave_mads_per_launch")
        except (OSError, KeyError, ValueError):
            pass
        mad_peak_file = SIMDS / MAD_NS             # the committed ubench figure: kept as a cross-check only
        live = calib is not None and "gmad_per_s" in calib
        mad_peak = calib["gmad_per_s"] if live else mad_peak_file
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl, "points_per_gpu": n, "window_bits": window_bits, "windows": plan["windows"],
                       "buckets": plan["buckets"], "entries_per_lane": plan["chunk"], "lanes": plan["lanes"],
                       "requests_in_flight": depth, "world_size": dist.get_world_size() if use_dist else 1,
                       "rccl_version": rccl_version,
                       "collective": coll_info.get("collective") if coll_info else None,
                       "collective_detail": coll_info,
                       "parallelism": "single GPU" if world == 1 else
                       (f"SRS-sharded x{world}, all_gather of 192 B partials" if is_msm
                        else f"Pianist segments x{world}, no exchange")},
            # `warmup` is the W the contract declares; what really preceded the timed region on this GPU is this count
            "msms_before_timed_region": msms_before + args.warmup,
            "msms_before_timed_region_note": "MSMs this GPU ran before the K timed steps: the `pipelined` measurement (taken first "
                                             "so that the one-at-a-time region starts on warm clocks) + the W declared warm-up steps",
            "identity": identity(),
            # what kzg_create measured about this process: lanes, lanes really running concurrently (hardware queues), HIP
            # already live when the library was loaded, GPU_MAX_HW_QUEUES as seen -- `pipelined` means little if this is < 2
            "runtime_info": eng.runtime_info(),
            "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel_ms": per_launch_s * 1e3 if acc_ms else None, "algorithmic_bytes": alg_bytes,
                         "measured_copy_peak_gbs": round(copy_gbs, 1) if copy_gbs else None,
                         "frac_of_measured_copy_peak": (achieved / (copy_gbs or HBM_COPY_GBS)) if achieved else None,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc, 2 x FETCH_SIZE + WRITE_SIZE)" if traffic else None,
                         "note": "integer-multiply-bound (14 x 28-bit-limb Montgomery products on v_mad_u64_u32), not "
                                 "HBM-bound: see mad_issue and DESIGN.md 3.3"},
            # what actually bounds the kernel: v_mad_u64_u32 wave-instructions per second against the
            # measured rate of that instruction (5.0 cycles per wave-instruction per SIMD, 1024 SIMDs, clock held under
            # this load); counts from the code object / SQ_INSTS_VALU passes under profiles/
            "mad_issue": ({"wave_mads_per_launch": mads, "wave_valu_insts_per_launch": valu_insts,
                           "achieved_gmad_s": mads / per_launch_s / 1e9, "peak_gmad_s": mad_peak,
                           "frac": mads / per_launch_s / 1e9 / mad_peak,
                           "peak_source": "kzg_calibrate in this run, right after the timed region" if live else
                                          "profiles/ubench_valu_rates.txt (another box: treat frac as +-14 %)",
                           "calibration": calib,
                           "peak_gmad_s_committed_ubench": mad_peak_file,
                           "live_vs_committed_peak": (mad_peak / mad_peak_file) if live else None,
                           "peak_basis": (f"{calib['simds']} SIMDs / {calib['ns_per_mad_per_simd']:.3f} ns per v_mad_u64_u32 "
                                          f"wave-instruction per SIMD at 2 waves per SIMD, measured in this run" if live else
                                          f"{SIMDS} SIMDs / {MAD_NS} ns per v_mad_u64_u32 wave-instruction "
                                          "(profiles/ubench_valu_rates.txt, 2 and 4 waves per SIMD)")
                                         + "; mads per launch from the ISA (profiles/isa_counts.json) x additions per lane "
                                           "x waves",
                           "simd32_full_rate_ginst_s": VALU_FULL_RATE_GINST_S}
                          if mads and acc_ms else None),
            "result_hex": results[0].hex() if isinstance(results[0], (bytes, bytearray)) else b"".join(results[0]).hex(),
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "stages_ms_source": f"{n_prof} extra steps with every stage bracketed by HIP events, after the timed region; "
                                "roofline.kernel_ms is measured inside the timed region"
                                + ("" if acc_live_ms else " of those extra steps (two-lane commit+open: timed unprofiled)"),
            "single_request_latency_ms": round(latency_ms, 4),
            "pipelined": pipelined,
            "step_ms": {"median": round(pctl(timed_step_ms, 0.5), 4), "p10": round(pctl(timed_step_ms, 0.1), 4),
                        "p90": round(pctl(timed_step_ms, 0.9), 4)},
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
        if world == 1 and args.workload == "msm20" and not args.no_adversarial and eng is not None:
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
        # ---- CPU baseline: the oracle's C port on a bounded sample of the same inputs (rank 0's segment when N > 1)
        threads = thread_counts(args.cpu_threads)
        if cpu_in is not None:
            from oracle import cpu as oc     # the checker / reported baseline: only this leg touches oracle/

            oc.build()
            threads = thread_counts(args.cpu_threads, oc.usable_cpus())
            m, srs, gpu_same = cpu_in
            where = "" if world == 1 else f" of rank 0's segment (the other {world - 1} rank(s) idle at a barrier meanwhile)"
            hw = (f"{host_cores()} host cores visible, {oc.usable_cpus()} usable under the cgroup quota, {cpu_model()}")
            if is_msm:
                per = {}
                cpu_res = None
                for th in threads:
                    mm = m if th > 1 else min(m, 1 << 18)          # one thread: 2^18 points (~1-2 s)
                    prep = oc.PreparedMsm(srs[: 96 * mm], scal[: 32 * mm])
                    tc = time.perf_counter()
                    r = prep.run(th)
                    per[th] = (mm, time.perf_counter() - tc)
                    prep.close()
                    if mm == m:
                        cpu_res = r
                if cpu_res is None:
                    cpu_res = oc.msm(srs, scal[: 32 * m], threads=max(threads))
                rates = {th: mm / s for th, (mm, s) in per.items()}
                best = max(rates, key=rates.get)
                out["cpu_baseline"] = {
                    "value": rates[best], "unit": "points/s", "cores": best, "kind": "port",
                    "sample": f"first 2^{m.bit_length() - 1} points/scalars of the same workload{where}, one Pippenger MSM "
                              f"split over the threads ({hw}); points/s by thread count: "
                              + ", ".join(f"{th}: {r:.0f} (2^{per[th][0].bit_length() - 1} pts, {per[th][1]:.2f} s)"
                                          for th, r in sorted(rates.items())),
                    "points_per_s_by_threads": {str(th): r for th, r in sorted(rates.items())},
                    "single_thread_points_per_s": rates.get(1),
                    "note": "plain-C stand-in (oracle/kzg_cpu.c, __int128 limbs); an asm-backed Pippenger (blst-class, what the reference's Rust prover most likely links) is expected to be 1.5-2x faster per core -- the GPU/CPU ratio is an upper estimate",
                    "matches_gpu_bit_exact": cpu_res == gpu_same,
                }
                assert cpu_res == gpu_same, "GPU MSM differs from the CPU oracle on the baseline sample"
            else:
                sample = scal[: 32 * m]
                per = {}
                cpu_res = None
                for th in threads:
                    if th == 1 and m > (1 << 14):
                        continue             # one thread on a long row would take minutes
                    tc = time.perf_counter()
                    c = oc.commit(srs, sample, True, threads=th)
                    ev, pf = oc.open_(srs, sample, alpha, True, threads=th)
                    per[th] = time.perf_counter() - tc
                    cpu_res = (c, ev, pf)
                best = min(per, key=per.get)
                out["cpu_baseline"] = {
                    "value": m / per[best], "unit": "coefficients/s", "cores": best, "kind": "port",
                    "sample": f"commit+open of the first 2^{m.bit_length() - 1} coefficients of the same row{where} "
                              f"(oracle/kzg_cpu.c; {hw}); seconds by thread count: "
                              + ", ".join(f"{th}: {s:.3f}" for th, s in sorted(per.items())),
                    "note": "plain-C stand-in (oracle/kzg_cpu.c, __int128 limbs); an asm-backed Pippenger (blst-class, what the reference's Rust prover most likely links) is expected to be 1.5-2x faster per core -- the GPU/CPU ratio is an upper estimate",
                    "matches_gpu_bit_exact": cpu_res == tuple(gpu_same)}
                assert cpu_res == tuple(gpu_same), "GPU commit+open differs from the CPU oracle on the baseline sample"
        if use_dist:
            # the headline is complete: print it NOW.  The extras below (the largest tables of the launch, a collective per
            # step) can then cost at most themselves; the augmented line printed after them supersedes this one (the LAST
            # JSON line is the record).
            flush_c_stdio()
            print(json.dumps(dict(out, partial_line="headline only: the extras of this launch follow in the next line")),
                  flush=True)
    if use_dist and args.workload == "msm20" and not args.no_dist_extra:
        try:
            dist_extra = dist_extra_workloads(args, ctl, eng, gather, lagrange_factor, rank, world)
        except BaseException as e:       # noqa: BLE001 -- belt and braces: the phases catch their own failures
            dist_extra = {"dist_extra_error": errmsg(e)}
            ctl.poisoned = ctl.poisoned or "an unexpected failure in the extra workloads"
    if rank == 0:
        if dist_extra:
            out.update(dist_extra)      # `pianist_kzg22` (configs[4]) and `msm26` (configs[3]) of the same launch
            if ctl.poisoned:
                out["process_group_note"] = ctl.poisoned
        if not use_dist:                # N = 1: free the headline's tables before the rows below build their own
            eng.close()
            eng = None
        if world == 1 and args.workload == "msm20" and not use_dist:
            # ---- KZG commit+open latency: the other half of BASELINE.json's metric (configs[2] + the production row sizes)
            if not args.no_kzg_rows:
                try:
                    logs = [int(x) for x in args.kzg_rows.split(",") if x]
                    out["kzg_commit_open"] = kzg_rows_report(HipEngine, lagrange_factor, local_rank, logs, threads,
                                                             not args.no_cpu_baseline)
                except Exception as e:   # noqa: BLE001 -- an extra never costs the headline
                    out["kzg_commit_open"] = {"error": errmsg(e)}
            # ---- the reference's own route and sizes, from text (VERDICT r4 task 5)
            if not args.no_e2e:
                try:
                    out["e2e_from_text"] = e2e_from_text_report(local_rank, [int(x) for x in args.e2e_rows.split(",") if x],
                                                                not args.no_cpu_baseline)
                except Exception as e:   # noqa: BLE001
                    out["e2e_from_text"] = {"error": errmsg(e)}
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    if leaked or ctl.poisoned:
        # a helper thread may still sit inside ncclCommInitRank, or a collective was abandoned half-way: the line is out --
        # leave without running destructors (communicator, process group) that would wait for peers that are gone
        flush_c_stdio()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if eng is not None:
        eng.close()
    if use_dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:                # noqa: BLE001 -- the line is out; a broken group must not turn into a failure code
            pass


if __name__ == "__main__":
    main()
