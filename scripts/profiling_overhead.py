#!/usr/bin/env python3
"""Step time of the 2^20 MSM through the blocking call and the ticket pair, at profiling levels 0 / 2 / 1 (what do the
HIP events inside the timed region of bench.py cost?)."""
import sys
import time

sys.path.insert(0, ".")
from bench import TAU, uniform_fr  # noqa: E402
from zkp_subnet_amd.engine import HipEngine  # noqa: E402

lg = 20
n = 1 << lg
sc = uniform_fr(n, seed=0)
eng = HipEngine(0)
eng.gen_srs(TAU, 1, lg, 0)
eng.upload_fr(0, sc, False)
for _ in range(30):
    eng.msm_resident(0, n, 0)


def run(label, fn, reps=150):
    for _ in range(10):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    print(f"{label:34s} {(time.perf_counter() - t) / reps * 1e3:.4f} ms/step", flush=True)


for rnd in range(2):
    for level in (0, 2, 1, 0):
        eng.set_profiling(level)
        run(f"blocking call, profiling {level}", lambda: eng.msm_resident(0, n, 0))
        run(f"submit + wait, profiling {level}", lambda: eng.msm_wait(eng.msm_submit(0, n, 0)))
