"""Helper of test_poly_kernel_variants_agree (run as a subprocess, with or without KZG_POLY_NO_LDS: the library reads
that A/B knob once per process): openings of long coefficient-form rows -- the sizes whose level-0 fold and quotient run
with 16 coefficients per lane -- against the C oracle, incl. alpha = 0, 1, a root of unity and r - 1, and eval alone."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bls12_381 as o            # noqa: E402
from oracle import cpu as oc                 # noqa: E402
from zkp_subnet_amd import HipEngine         # noqa: E402

for lg in [int(a) for a in sys.argv[1:]]:
    T = 1 << lg
    eng = HipEngine(0)
    eng.gen_srs(0xABCDEF + lg, 1, lg, 0)
    srs = eng.srs_read(0, T)
    raw = np.random.default_rng(lg).integers(0, 256, size=(T, 32), dtype=np.uint8)
    raw[:, 0] &= 0x3F
    row = raw.tobytes()
    omega = pow(7, (o.R - 1) // T, o.R)
    for a in (0x1234567890ABCDEF1234567890ABCDEF % o.R, 0, 1, pow(omega, 12345, o.R), o.R - 1):
        alpha = a.to_bytes(32, "big")
        assert eng.open(0, row, alpha, False) == oc.open_(srs, row, alpha, False, threads=8), ("open", lg, hex(a))
        assert eng.eval(row, alpha) == oc.fr_eval(row, alpha), ("eval", lg, hex(a))
    # eval of a coefficient vector whose length is NOT a power of two: a multiple of 1024 (the staged form's granularity)
    # and one that is not (falls back to the strided form whatever the knob says)
    for n in (T + 3 * 1024, T + 3 * 1024 + 7):
        ext = (row * 2)[:32 * n]
        alpha = (0xFEDCBA9876543210 % o.R).to_bytes(32, "big")
        assert eng.eval(ext, alpha) == oc.fr_eval(ext, alpha), ("eval ragged", lg, n)
    eng.close()
print("ok")
