"""NOTE (round 6): the experiment is CLOSED and its hook (KZG_EXP_L2_RESIDENT in k_msm_accumulate) no longer lives in the shipped
kernel; `git apply -R scripts/proto/exp_l2_resident.removed.patch` puts it back for a re-run.

VERDICT r4 task 6: does the 16.5 x HBM traffic of k_msm_accumulate (2.22 GB gathered per 2^20-point launch, ~1.1 TB/s)
cost the kernel CLOCK (socket power cap) or TIME?  One experiment, same box, same process layout:

  real    the shipped library: every digit gathers its own 128-byte row from one of 13 window tables (1.74 GB resident)
  l2      a TIMING build (-DKZG_EXP_L2_RESIDENT, zkp_subnet_amd/ab/L2.so): every row index is masked to the first 2^14
          points of window table 0 (2 MB: resident in L2 / Infinity Cache) -- same instruction stream, same mads, same
          sorted-index reads and bucket stores, ~no HBM gather; the results are garbage and are not looked at
  l2q     -DKZG_EXP_L2_RESIDENT=25 (ab/L2q.so): only every fourth row is redirected -- a quarter of the gather gone, what a
          denser row (2 x 48 B packed into 96 B) could save at best, priced BEFORE paying for its unpacking

For each: ~3 s of back-to-back 2^20 MSMs with HIP events around the accumulate kernel only (profiling level 2), the socket
power sampled from sysfs every 50 ms meanwhile (when the container exposes it), then kzg_calibrate (s_memtime ticks per ns
= the clock the SIMDs ran at right after that load).  Usage (GPU box):  python scripts/exp_traffic_clock.py
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def power_nodes():
    """The power sensor of THE GPU THIS PROCESS USES (the box's sysfs shows all eight cards of the host): matched by the PCI
    address HIP reports for device 0; every card's node if that cannot be resolved (then only the MAX is meaningful)."""
    nodes = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") +
                   glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
    try:
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            bdf = buf.value.decode().lower()
            mine = [p for p in nodes if bdf in os.path.realpath(os.path.join(os.path.dirname(p), "..", "..")).lower()]
            if mine:
                return mine
    except OSError:
        pass
    return nodes


def read_power(nodes):
    vals = []
    for p in nodes:
        try:
            with open(p) as f:
                vals.append(int(f.read().strip()) / 1e6)      # microwatts -> W
        except (OSError, ValueError):
            pass
    return max(vals) if vals else None


def child(tag):
    from bench import TAU, uniform_fr
    from zkp_subnet_amd import HipEngine

    lg = 20
    n = 1 << lg
    eng = HipEngine(0)
    eng.gen_srs(TAU, 1, lg, 0)
    eng.upload_fr(0, uniform_fr(n, seed=0), False)
    for _ in range(30):                       # ~80 ms: the clocks need ~40 ms of load
        eng.msm_resident(0, n, 0)
    nodes = power_nodes()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            w = read_power(nodes)
            if w is not None:
                samples.append(w)
            time.sleep(0.05)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    eng.set_profiling(2)
    acc, steps = [], 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        eng.msm_resident(0, n, 0)
        acc.append(eng.timings()["accumulate"])
        steps += 1
    wall = time.perf_counter() - t0
    eng.set_profiling(0)
    cal = eng.calibrate(2)
    stop.set()
    th.join()
    acc.sort()
    rec = {"variant": tag, "lib": os.environ.get("KZG_MI355X_LIB", "shipped"), "msm_per_s": steps / wall, "ms_per_msm": wall / steps * 1e3,
           "accumulate_ms_median": acc[len(acc) // 2], "accumulate_ms_p10": acc[len(acc) // 10], "accumulate_ms_p90": acc[9 * len(acc) // 10],
           "memtime_ticks_per_ns_after": cal["memtime_ticks_per_ns"], "ns_per_mad_per_simd_after": cal["ns_per_mad_per_simd"],
           "power_w": ({"samples": len(samples), "mean": sum(samples) / len(samples), "max": max(samples), "min": min(samples)}
                       if samples else None), "power_nodes": nodes}
    eng.close()
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    l2 = os.path.join(ROOT, "zkp_subnet_amd", "ab", "L2.so")
    if not os.path.exists(l2):
        sys.exit("build the timing library first: KZG_BUILD_TAG=L2 KZG_EXTRA_HIPCC_FLAGS=-DKZG_EXP_L2_RESIDENT python -m zkp_subnet_amd.build")
    for rnd in range(3):                      # interleaved: a drifting box shows up as drift in BOTH columns
        for tag, lib in (("real", None), ("l2", l2), ("l2q", os.path.join(ROOT, "zkp_subnet_amd", "ab", "L2q.so"))):
            if lib and not os.path.exists(lib):
                continue
            env = dict(os.environ)
            env.pop("KZG_MI355X_LIB", None)
            if lib:
                env["KZG_MI355X_LIB"] = lib
            subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=env, check=False)
    # what an idle socket draws, for scale
    nodes = power_nodes()
    time.sleep(1.0)
    print(json.dumps({"idle_power_w": read_power(nodes), "power_nodes": nodes}))
