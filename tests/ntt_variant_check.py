"""Helper of test_ntt_kernel_variants_agree (run as a subprocess with KZG_NTT_RADIX2 / KZG_NTT_TILE_LOG set: the library
reads those A/B knobs once per process): forward and inverse transforms of a few sizes against the C oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu as oc                 # noqa: E402
from zkp_subnet_amd import HipEngine         # noqa: E402

eng = HipEngine(0)
for lg in [int(a) for a in sys.argv[1:]]:
    raw = np.random.default_rng(lg).integers(0, 256, size=(1 << lg, 32), dtype=np.uint8)
    raw[:, 0] &= 0x3F
    v = raw.tobytes()
    assert eng.ntt(v, False) == oc.fr_ntt(v, False), ("forward", lg)
    assert eng.ntt(v, True) == oc.fr_ntt(v, True), ("inverse", lg)
eng.close()
print("ok")
