"""A few resident 2^20-point MSMs, for a rocprofv3 --kernel-trace of the tail kernels (scripts/tree_levels.py reads it).
Errors of the result decoding are ignored: the timing-experiment builds (-DKZG_TREE_EXP) compute garbage on purpose."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from zkp_subnet_amd import HipEngine  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = HipEngine(0, window=0)
eng.gen_srs(5, 7, lg, 0, [0])
n = 1 << lg
sc = np.random.default_rng(1).integers(0, 256, size=32 * n, dtype=np.uint8)
sc[::32] &= 0x3F
eng.upload_fr(1, sc.tobytes(), False)
bad = 0
for _ in range(12):
    try:
        eng.msm_resident(1, n, 0)
    except Exception:
        bad += 1
print("done, decode errors:", bad)
