"""Child process of test_gpu_multi.py's device-count-gated tests: ONE real rank of an N-rank job, one GPU per rank, NO torch in
the process.  The rendezvous is a file (rank 0 writes the 128-byte RCCL unique id, the others poll for it): the library's own
collective needs nothing else (`kzg_comm_unique_id` -> `kzg_comm_init_bounded` -> `kzg_comm_selftest` -> `kzg_msm_sharded`).
Rank g holds SRS segment g (points [g n, (g + 1) n) of the flat SRS [tau^j] G) and the scalars of the same index range; every
rank prints the 48-byte result of each sharded MSM, which the parent compares with the oracle's trapdoor value [f(tau)] G.

    python tests/comm_ranks.py <rank> <world> <dir> <tau hex> <init_timeout_ms> <lg,lg,...> [absent]

`absent`: this job is deliberately one rank short -- the ranks that did start must get KZG_E_COMM inside the budget and keep
a working engine (a plain MSM afterwards still equals the trapdoor value)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")        # one node: RCCL's bootstrap over loopback (the hostname may not resolve)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks' processes (this pool's host driver)
import numpy as np  # noqa: E402

from zkp_subnet_amd import HipEngine, KzgError  # noqa: E402
from zkp_subnet_amd._native import KZG_E_COMM  # noqa: E402
from zkp_subnet_amd.engine import R_MODULUS  # noqa: E402

rank, world, d, tau, init_ms = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4], 16), int(sys.argv[5])
logs = [int(x) for x in sys.argv[6].split(",")]
absent = len(sys.argv) > 7 and sys.argv[7] == "absent"
assert "torch" not in sys.modules


def scalars(lg, r):
    raw = np.random.default_rng(1000 * lg + r).integers(0, 256, size=(1 << lg, 32), dtype=np.uint8)
    raw[:, 0] &= 0x3F
    return raw.tobytes()


eng = HipEngine(rank)                       # one GPU per rank
id_path = os.path.join(d, "id.bin")
if rank == 0:
    with open(id_path + ".tmp", "wb") as f:
        f.write(HipEngine.comm_unique_id())
    os.rename(id_path + ".tmp", id_path)
t0 = time.time()
while not os.path.exists(id_path):
    if time.time() - t0 > 120:
        sys.exit("rank 0 never published the unique id")
    time.sleep(0.01)
with open(id_path, "rb") as f:
    uid = f.read()
out = {"rank": rank, "runtime": eng.runtime_info()}
t0 = time.perf_counter()
try:
    eng.comm_init(uid, rank, world, timeout_ms=60000, init_timeout_ms=init_ms)
    out["init"] = "ok"
except KzgError as e:
    out["init"] = "E_COMM" if e.code == KZG_E_COMM else f"code {e.code}"
    out["init_error"] = str(e)[:300]
out["init_s"] = round(time.perf_counter() - t0, 3)
if out["init"] == "ok":
    eng.comm_selftest()
    out["info"] = eng.comm_info()
    for lg in logs:
        n = 1 << lg
        eng.gen_srs(tau, 1, lg, 0, factors=[pow(tau, rank * n, R_MODULUS)])
        eng.upload_fr(0, scalars(lg, rank), False)
        res = [eng.msm_sharded(0, n, 0).hex() for _ in range(2)]
        out[f"msm_{lg}"] = res[0] if res[0] == res[1] else "UNSTABLE"
        eng.set_profiling(1)
        eng.msm_sharded(0, n, 0)
        out[f"collective_ms_{lg}"] = round(eng.timings()["collective"], 4)
        eng.set_profiling(0)
    eng.comm_destroy()
if absent or out["init"] != "ok":
    # whatever happened to the communicator, the engine itself must still serve
    lg = logs[0]
    eng.gen_srs(tau, 1, lg, 0)
    out["plain_after"] = eng.msm(scalars(lg, rank), 0).hex()
eng.close()
print(json.dumps(out), flush=True)
if out["init"] != "ok":
    os._exit(0)      # a join thread may still sit inside RCCL's rendezvous: the line is out, leave without its destructors
