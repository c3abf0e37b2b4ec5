"""Stage timings of one 2^LOG_N MSM for adversarial scalar distributions (dev tool; needs the GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import uniform_fr, TAU
from zkp_subnet_amd import HipEngine

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
eng = HipEngine(0)
eng.gen_srs(TAU, 1, lg, 0)
small = np.zeros((n, 32), dtype=np.uint8)
small[:, 28:] = np.random.default_rng(78).integers(0, 256, size=(n, 4), dtype=np.uint8)
two = (uniform_fr(1, 5) * (n // 2)) + (uniform_fr(1, 6) * (n // 2))
cases = {"uniform": uniform_fr(n, 0), "all_equal": uniform_fr(1, 77) * n, "two_values": two, "below_2^32": small.tobytes()}
eng.set_profiling(True)
for name, data in cases.items():
    eng.upload_fr(1, data, False)
    eng.msm_resident(1, n, 0)
    eng.msm_resident(1, n, 0)
    t = eng.timings()
    print(name, {k: round(v, 3) for k, v in t.items() if v})
