#!/usr/bin/env python3
"""Per-step time of consecutive 2^20 MSMs from a cold start: does the GPU need sustained load to reach its clocks?
(development aid; bench.py's K timed steps follow W warm-up steps, the driver uses W = 5)"""
import sys
import time

sys.path.insert(0, ".")
from bench import TAU, uniform_fr  # noqa: E402
from zkp_subnet_amd.engine import HipEngine  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = HipEngine(0)
eng.gen_srs(TAU, 1, lg, 0)
eng.upload_fr(0, uniform_fr(1 << lg, seed=0), False)
time.sleep(float(sys.argv[2]) if len(sys.argv) > 2 else 2.0)   # let the GPU go idle after the table build
ts = []
for i in range(400):
    t = time.perf_counter()
    eng.msm_resident(0, 1 << lg, 0)
    ts.append((time.perf_counter() - t) * 1e3)
for i in list(range(0, 20)) + list(range(20, 400, 20)):
    print(f"step {i:3d}: {ts[i]:.3f} ms   (elapsed {sum(ts[:i]):7.1f} ms)")
