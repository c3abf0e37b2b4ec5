"""Child process of test_gpu_multi.py: the library's OWN collective (kzg_comm_init / kzg_msm_sharded) on a one-rank
communicator on cuda:0, with NO torch in the process (the reference seam is one client object with no framework
underneath, base/miner.py:73-84).  The sharded MSM -- partial, ncclAllGather on the lane's stream, sum -- must return
the bytes of the plain single-GPU MSM, from several host threads at once, after a timeout that aborts the communicator,
and after the communicator has been rebuilt.  Prints one JSON line."""
import ctypes
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from zkp_subnet_amd import HipEngine, KzgError  # noqa: E402
from zkp_subnet_amd._native import KZG_E_ARG, KZG_E_COMM  # noqa: E402

assert "torch" not in sys.modules
out = {}
for lg in (6, 12, 16, 20):      # ... up to the headline size (BASELINE.json configs[1])
    n = 1 << lg
    eng = HipEngine(0)
    eng.gen_srs(0x51AB1E + lg, 1, lg, 0)
    raw = np.random.default_rng(lg).integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw[:, 0] &= 0x3F
    eng.upload_fr(0, raw.tobytes(), False)
    plain = eng.msm_resident(0, n, 0)
    rec = {"plain": plain.hex()}
    try:
        eng.msm_sharded(0, n, 0)
        rec["without_comm"] = "no error"
    except KzgError as e:
        rec["without_comm"] = e.code == KZG_E_ARG
    eng.comm_init(HipEngine.comm_unique_id(), 0, 1)
    eng.comm_selftest()                     # a checked all_gather: what bench.py's preflight runs before building tables
    rec["info"] = eng.comm_info()
    rec["sharded_equal"] = all(eng.msm_sharded(0, n, 0) == plain for _ in range(3))
    rec["segment_equal"] = eng.msm_sharded(0, n // 2, n // 4) == eng.msm_resident(0, n // 2, n // 4)
    try:
        eng.comm_init(HipEngine.comm_unique_id(), 0, 1)
        rec["double_init"] = "no error"
    except KzgError as e:
        rec["double_init"] = e.code == KZG_E_ARG
    eng.set_profiling(1)
    assert eng.msm_sharded(0, n, 0) == plain
    rec["collective_ms"] = eng.timings()["collective"]
    eng.set_profiling(0)
    # four host threads, each on its own lane, sharing the one communicator
    bad = []

    def worker():
        for _ in range(6):
            if eng.msm_sharded(0, n, 0) != plain:
                bad.append(1)

    th = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    rec["threads_equal"] = not bad
    if lg == 12:
        # a "peer" that is 400 ms late against a 100-ms budget: KZG_E_COMM, the communicator aborted, later calls refused
        # at once, the rest of the library unharmed; then a fresh communicator
        eng.comm_set_timeout(100)
        assert eng.msm_sharded(0, n, 0) == plain                      # the bounded wait itself, nothing late
        eng._chk(eng._lib.kzg_test_comm_stall(eng._h, 400))
        t0 = time.perf_counter()
        try:
            eng.msm_sharded(0, n, 0)
            rec["timeout"] = "no error"
        except KzgError as e:
            rec["timeout"] = e.code == KZG_E_COMM
        rec["timeout_after_ms"] = (time.perf_counter() - t0) * 1e3
        rec["broken"] = eng.comm_info()["broken"]
        try:
            eng.msm_sharded(0, n, 0)
            rec["after_abort"] = "no error"
        except KzgError as e:
            rec["after_abort"] = e.code == KZG_E_COMM
        rec["plain_after_abort"] = eng.msm_resident(0, n, 0) == plain
        eng.comm_destroy()
        eng.comm_init(HipEngine.comm_unique_id(), 0, 1, timeout_ms=30000)
        rec["rebuilt_equal"] = eng.msm_sharded(0, n, 0) == plain and not eng.comm_info()["broken"]
    eng.comm_destroy()
    rec["world_after_destroy"] = eng.comm_info()["world"]
    out[str(lg)] = rec
    eng.close()
ctypes.CDLL(None).fflush(None)
print(json.dumps(out))
