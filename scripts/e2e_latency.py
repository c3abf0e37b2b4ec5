"""End-to-end latency of the drop-in boundary on one MI355X: Client.worker_commit_and_open(i, poly: List[str], x: str)
from the reference's wire form (43-char base64 strings) to the response strings, i.e. text codec + H2D + GPU + D2H.
Prints one JSON line per size with the share of each part (dev tool; needs the GPU)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import uniform_fr                                   # noqa: E402
from zkp_subnet_amd import codec                               # noqa: E402
from zkp_subnet_amd.client import Client                       # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [12, 16, 20, 22]
for lg in sizes:
    T = 1 << lg
    cl = Client(seed=3, workers=[0])
    cl.start(scale=lg, machines_scale=0)
    row = uniform_fr(T, 1)
    poly = codec.be32_to_fr_list(row)
    x = codec.be32_to_fr(uniform_fr(1, 2))
    t_w = time.perf_counter()
    warm = 0
    while warm < 2 or time.perf_counter() - t_w < 0.08:      # the GPU needs ~40 ms of load to reach its clocks
        with cl.worker_commit_and_open(0, poly, x) as r:
            assert r.status_code == 200, r.json()
        warm += 1
    reps = 5 if lg >= 20 else 40
    t0 = time.perf_counter()
    for _ in range(reps):
        with cl.worker_commit_and_open(0, poly, x) as r:
            body = r.json()
    e2e = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        raw = codec.fr_list_to_be32(poly)
    dec = (time.perf_counter() - t0) / reps
    # what the Client path actually pays: the same decode straight into one of the library's pinned staging buffers (no
    # 32 T-byte bytes object to allocate and fault in -- at 2^22 that allocation is most of `text_decode_ms`)
    from zkp_subnet_amd.engine import HipEngine               # noqa: E402
    t0 = time.perf_counter()
    for _ in range(reps):
        with HipEngine._Staged(cl.engine, poly):
            pass
    dec_pinned = (time.perf_counter() - t0) / reps
    xb = codec.fr_to_be32(x)
    t0 = time.perf_counter()
    for _ in range(reps):
        cl.engine.commit_open(0, raw, xb, True)
    host_buf = (time.perf_counter() - t0) / reps
    cl.engine.upload_fr(0, raw, True)
    t0 = time.perf_counter()
    for _ in range(reps):
        cl.engine.commit_open_resident(0, 0, T, xb, True)
    resident = (time.perf_counter() - t0) / reps
    # the reference's own route for comparison: two calls, poly shipped and decoded twice (neurons/miner.py:56-61)
    t0 = time.perf_counter()
    for _ in range(reps):
        with cl.worker_commit(0, poly) as a, cl.worker_open(0, poly, x) as b:
            assert a.json()["commitment"] == body["commitment"] and b.json()["proof"] == body["proof"]
    two_call = (time.perf_counter() - t0) / reps
    hits, misses = cl.engine.row_cache_stats()
    print(json.dumps({"log2_T": lg, "row_cache_hits_misses": [hits, misses], "e2e_fused_ms": round(e2e * 1e3, 3), "text_decode_ms": round(dec * 1e3, 3), "text_decode_into_pinned_ms": round(dec_pinned * 1e3, 3),
                      "host_buffer_call_ms": round(host_buf * 1e3, 3), "resident_call_ms": round(resident * 1e3, 3),
                      "two_call_route_ms": round(two_call * 1e3, 3), "wire_ext": codec._wire is not None,
                      "streamed_upload": T >= HipEngine.STREAM_MIN}), flush=True)
    cl.stop()
