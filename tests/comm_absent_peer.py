"""Child process of test_gpu_multi.py: a communicator whose peer NEVER arrives, on one GPU, no torch.  world = 2 but only rank 0
calls kzg_comm_init_bounded: the rendezvous runs on the library's helper thread, the call returns KZG_E_COMM inside the
budget, and the context is NOT wedged -- plain MSMs, a fresh one-rank communicator with sharded MSMs on the same engine,
engine.close() and a normal interpreter exit all still work (round 5's Python-side watchdog left the engine holding
every lane and comm.mu).  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from zkp_subnet_amd import HipEngine, KzgError  # noqa: E402
from zkp_subnet_amd._native import KZG_E_COMM  # noqa: E402

lg, tau = 12, 0xAB5E47
n = 1 << lg
raw = np.random.default_rng(77).integers(0, 256, size=(n, 32), dtype=np.uint8)
raw[:, 0] &= 0x3F
eng = HipEngine(0)
eng.gen_srs(tau, 1, lg, 0)
eng.upload_fr(0, raw.tobytes(), False)
out = {"plain": eng.msm_resident(0, n, 0).hex()}
t0 = time.perf_counter()
try:
    eng.comm_init(HipEngine.comm_unique_id(), 0, 2, init_timeout_ms=1500)
    out["absent_peer"] = "no error"
except KzgError as e:
    out["absent_peer"] = e.code == KZG_E_COMM
    out["message"] = str(e)[:200]
out["gave_up_after_s"] = round(time.perf_counter() - t0, 3)
out["info_after"] = eng.comm_info()
out["plain_after"] = eng.msm_resident(0, n, 0).hex()
# a NEW communicator on the same engine while the abandoned join thread still sits in RCCL's rendezvous
eng.comm_init(HipEngine.comm_unique_id(), 0, 1, timeout_ms=30000, init_timeout_ms=60000)
out["sharded_after"] = eng.msm_sharded(0, n, 0).hex()
eng.comm_destroy()
eng.close()
out["closed"] = True
print(json.dumps(out), flush=True)
