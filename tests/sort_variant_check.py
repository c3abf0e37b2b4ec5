"""Helper of test_sort_round_variants_agree (run as a subprocess with KZG_SORT_ROUNDS set: the library reads that A/B knob
once per process): MSMs whose level-1 partition runs with the forced number of rounds per workgroup, against the C oracle --
well-spread, skewed (region overflow -> exact mode), all-equal and tiny scalars, ragged lengths, a batch of two scalar sets
(commit+open), windows of 16 .. 24 bits (<= 16 windows: the two-round kernel's precondition)."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bls12_381 as o            # noqa: E402
from oracle import cpu as oc                 # noqa: E402
from zkp_subnet_amd import HipEngine         # noqa: E402

rnd = random.Random(int(os.environ.get("KZG_SORT_ROUNDS", "0")) + 77)
oc.build()


def scalars(n, kind):
    if kind == "uniform":
        return b"".join(rnd.randrange(o.R).to_bytes(32, "big") for _ in range(n))
    if kind == "small":
        return b"".join(rnd.randrange(1 << 33).to_bytes(32, "big") for _ in range(n))
    if kind == "equal":
        return rnd.randrange(1, o.R).to_bytes(32, "big") * n
    base = rnd.randrange(o.R >> 1) & ~((1 << 40) - 1)   # clustered: a few adjacent buckets take everything
    return b"".join((base + rnd.randrange(1 << 9)).to_bytes(32, "big") for _ in range(n))


for lg, window in ((13, 16), (14, 20), (15, 17), (12, 24), (16, 0)):
    eng = HipEngine(0, window=window)
    tx, ty = rnd.randrange(2, o.R), rnd.randrange(2, o.R)
    eng.gen_srs(tx, ty, lg, 0, [0])
    T = 1 << lg
    srs = oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), lg, 0, 0)
    assert eng.srs_read(0, T) == srs
    for kind in ("uniform", "clustered", "equal", "small"):
        n = T if kind == "uniform" else rnd.randrange(T // 2, T + 1)
        off = T - n
        sc = scalars(n, kind)
        assert eng.msm(sc, off) == oc.msm(srs[96 * off:96 * (off + n)], sc), (lg, window, kind, n)
    row = scalars(T, "uniform")
    alpha = rnd.randrange(o.R).to_bytes(32, "big")
    want = (oc.commit(srs, row, True, threads=8),) + tuple(oc.open_(srs, row, alpha, True, threads=8))
    assert eng.commit_open(0, row, alpha, True) == want, ("commit_open", lg, window)
    eng.close()
print("ok")
