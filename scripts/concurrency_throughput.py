"""Throughput of concurrent host threads through the reference-facing surface (dev tool; needs the GPU): K threads each
call Client.worker_commit_and_open(i, List[str], str) in a loop on ONE Client / ONE context, as the reference's axon
does with Miner.forward (neurons/miner.py:106-135).  Each call runs on its own lane of the context, so one request's
sort and latency-bound tail hide under another's accumulate.  Prints one JSON line per row length.

    python scripts/concurrency_throughput.py [log2_T ...]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import uniform_fr                                   # noqa: E402
from zkp_subnet_amd import codec                               # noqa: E402
from zkp_subnet_amd.client import Client                       # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [12, 16, 20]
for lg in sizes:
    T = 1 << lg
    cl = Client(seed=3, workers=[0])
    cl.start(scale=lg, machines_scale=0)
    polys = [codec.be32_to_fr_list(uniform_fr(T, 10 + k)) for k in range(8)]
    x = codec.be32_to_fr(uniform_fr(1, 2))
    want = []
    for p in polys:
        with cl.worker_commit_and_open(0, p, x) as r:
            assert r.status_code == 200, r.json()
            want.append(r.json())
    reps = 24 if lg <= 16 else 8
    out = {"log2_T": lg}
    for threads in (8, 1, 2, 4, 8):       # the first pass only warms every lane's workspace
        bad = []

        def work(tid):
            for it in range(reps):
                k = (tid + it) % 8
                with cl.worker_commit_and_open(0, polys[k], x) as r:
                    if r.json() != want[k]:
                        bad.append((tid, it))

        ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        assert not bad, bad
        out[f"requests_per_s_{threads}_threads"] = round(threads * reps / dt, 1)
        out[f"ms_per_request_{threads}_threads"] = round(dt / (threads * reps) * 1e3, 3)
    out["speedup_2_threads"] = round(out["requests_per_s_2_threads"] / out["requests_per_s_1_threads"], 3)
    out["speedup_4_threads"] = round(out["requests_per_s_4_threads"] / out["requests_per_s_1_threads"], 3)
    print(json.dumps(out), flush=True)
    cl.stop()
