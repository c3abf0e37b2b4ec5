"""Which clock state is this box in, and does sustained load change it?  Logs kzg_calibrate (ns per v_mad_u64_u32
wave-instruction per SIMD, s_memtime ticks per ns) every ~0.25 s for `seconds` of back-to-back 2^20 MSMs from process
start (dev tool; needs the GPU).  Boxes of the pool were seen at 2.29 ns / 2.40 GHz and at 2.50 ns / 2.20 GHz, one box in
both states in two consecutive processes (profiles/r04_bench_default.json vs r04_bench_driver_style.json).

    python scripts/clock_state.py [seconds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import TAU, uniform_fr, sysfs_sclk_mhz              # noqa: E402
from zkp_subnet_amd import HipEngine                           # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
t_proc = time.perf_counter()
eng = HipEngine(0)
first = eng.calibrate(2)                                       # the very first GPU work of the process
print(json.dumps({"t_s": round(time.perf_counter() - t_proc, 3), "phase": "first call", "ns_per_mad": round(first["ns_per_mad_per_simd"], 4),
                  "ticks_per_ns": round(first["memtime_ticks_per_ns"], 4)}), flush=True)
eng.gen_srs(TAU, 1, 20, 0)
eng.upload_fr(0, uniform_fr(1 << 20, 0), False)
t0 = time.perf_counter()
nxt = 0.0
while time.perf_counter() - t0 < seconds:
    ts = time.perf_counter()
    n = 0
    while time.perf_counter() - ts < 0.25:
        eng.msm_resident(0, 1 << 20, 0)
        n += 1
    ms = (time.perf_counter() - ts) / n * 1e3
    c = eng.calibrate(2)
    print(json.dumps({"t_s": round(time.perf_counter() - t_proc, 3), "msm20_ms": round(ms, 4), "ns_per_mad": round(c["ns_per_mad_per_simd"], 4),
                      "ticks_per_ns": round(c["memtime_ticks_per_ns"], 4), "sclk_sysfs": sysfs_sclk_mhz(0)}), flush=True)
eng.close()
