"""Thread-stress drive of the HOST-side native code for the sanitizer builds (scripts/sanitize_cpu.sh): not collected
by pytest, needs no GPU.  Eight Python threads (the shape of the reference's axon worker threads,
neurons/miner.py:106-135) call, concurrently and repeatedly:
  - the wire codec's worker pool (csrc/wire_py.c): decode (plain / into a buffer / tagged), encode, and the
    asynchronous-batch path behind random_fr_rows;
  - kzg_vk_verify / kzg_vk_verify_batch (csrc/pairing_host.cpp): per-call std::thread pool;
  - the oracle's task pool (oracle/kzg_cpu.c: threaded MSM, commit, open) -- test infrastructure, but its races would
    make the parity tests lie.
Every answer is checked against a single-threaded one computed up front.  `python tests/san_drive.py [iterations]`"""
import ctypes
import os
import random
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bls12_381 as o                     # noqa: E402
from oracle import cpu as oc                          # noqa: E402
from zkp_subnet_amd import codec                      # noqa: E402
from zkp_subnet_amd.verifier import Verifier          # noqa: E402

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 6
NTHREADS = 8
assert codec._wire is not None, "zkp_subnet_amd._wire is not built"
w = codec._wire
rnd = random.Random(1)
# ---- answer book (single-threaded)
rows = {n: b"".join(rnd.randrange(o.R).to_bytes(32, "big") for _ in range(n)) for n in (1, 700, 3000, 40000)}
polys = {n: w.encode_fr_list(r) for n, r in rows.items()}
for n, r in rows.items():
    assert w.decode_fr_list(polys[n], 1) == r
tags = {}
for n, r in rows.items():
    buf = ctypes.create_string_buffer(len(r))
    got, tags[n] = w.decode_fr_list_into_tagged(polys[n], ctypes.addressof(buf), len(r))
    assert got == n and buf.raw == r
lg, ms = 7, 1
T = 1 << (lg - ms)
tx, ty = 0x1234ABCD, 0x7777
srs = [oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), lg, ms, i) for i in range(2)]
from zkp_subnet_amd.engine import lagrange_factor     # noqa: E402

vk = Verifier.synthetic(tx, [lagrange_factor(i, ms, ty) for i in range(2)])
kz_rows = [b"".join(rnd.randrange(o.R).to_bytes(32, "big") for _ in range(T)) for _ in range(4)]
alpha = rnd.randrange(o.R).to_bytes(32, "big")
book = []
for k, row in enumerate(kz_rows):
    i = k & 1
    c = oc.commit(srs[i], row, True, threads=1)
    ev, pf = oc.open_(srs[i], row, alpha, True, threads=1)
    book.append((i, c, ev, pf))
    assert vk.verify(i, pf, alpha, ev, c)
msm_sc = b"".join(rnd.randrange(o.R).to_bytes(32, "big") for _ in range(T))
msm_want = oc.msm(srs[0], msm_sc, threads=1)
errors = []


def worker(tid):
    r = random.Random(100 + tid)
    try:
        for it in range(ITER):
            n = r.choice((1, 700, 3000, 40000))
            if w.decode_fr_list(polys[n], r.choice((0, 2, 8, 16))) != rows[n]:
                errors.append((tid, "decode", n))
            buf = ctypes.create_string_buffer(32 * n)
            got, tag = w.decode_fr_list_into_tagged(polys[n], ctypes.addressof(buf), 32 * n)
            if got != n or buf.raw != rows[n] or tag != tags[n]:
                errors.append((tid, "tagged", n))
            if w.decode_fr_list_into(polys[n], ctypes.addressof(buf), 32 * n) != n or buf.raw != rows[n]:
                errors.append((tid, "into", n))
            if n >= 700:                                       # the tile-by-tile form of long rows: ranges + tag shares
                part = bytes(16)
                tile = n // 3 + 1
                for first in range(0, n, tile):
                    k, part = w.decode_fr_list_into_tagged(polys[n], ctypes.addressof(buf), 32 * n, r.choice((0, 4, 16)),
                                                           first, min(tile, n - first), part)
                if part != tags[n] or buf.raw != rows[n]:
                    errors.append((tid, "ranges", n))
            if w.encode_fr_list(rows[n]) != polys[n]:
                errors.append((tid, "encode", n))
            rr = w.random_fr_rows(r.choice((1, 3)), r.choice((5, 2000, 9000)))      # the asynchronous-batch path
            flat = [s for row in rr for s in row]
            if any(int.from_bytes(x, "big") >= o.R for x in (lambda b: [b[i:i + 32] for i in range(0, len(b), 32)])(w.decode_fr_list(flat, 4))):
                errors.append((tid, "random_fr_rows"))
            i, c, ev, pf = book[r.randrange(4)]
            if not vk.verify(i, pf, alpha, ev, c):
                errors.append((tid, "verify"))
            idx = [b[0] for b in book]
            if not vk.verify_batch(idx, [b[3] for b in book], alpha, [b[2] for b in book], [b[1] for b in book], threads=8):
                errors.append((tid, "verify_batch"))
            bad = [b[3] for b in book]
            bad[1], bad[3] = bad[3], bad[1]
            if vk.verify_batch(idx, bad, alpha, [b[2] for b in book], [b[1] for b in book], threads=3):
                errors.append((tid, "verify_batch accepted swapped proofs"))
            if oc.msm(srs[0], msm_sc, threads=r.choice((2, 4, 8))) != msm_want:
                errors.append((tid, "oracle msm"))
            k = r.randrange(4)
            i, c, ev, pf = book[k]
            if oc.commit(srs[i], kz_rows[k], True, threads=4) != c or oc.open_(srs[i], kz_rows[k], alpha, True, threads=4) != (ev, pf):
                errors.append((tid, "oracle commit/open"))
    except Exception as e:      # noqa: BLE001
        errors.append((tid, repr(e)))


threads = [threading.Thread(target=worker, args=(t,)) for t in range(NTHREADS)]
for t in threads:
    t.start()
for t in threads:
    t.join()
assert not errors, errors[:10]
print(f"san_drive ok: {NTHREADS} threads x {ITER} iterations, every answer as single-threaded")
