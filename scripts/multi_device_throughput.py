"""Requests per second of a miner host's Pianist rows through MultiDeviceClient with 1, 2 (and more) contexts (dev tool;
needs the GPU).  `--devices 0,0` puts two contexts on ONE GPU (what a one-GPU box can measure: a second context adds
lanes and a second set of workspaces, not SIMDs); on a multi-GPU host pass distinct devices.

    python scripts/multi_device_throughput.py [--devices 0,0] [log2_T ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import uniform_fr                                   # noqa: E402
from zkp_subnet_amd import MultiDeviceClient, codec            # noqa: E402

args = sys.argv[1:]
devs = [0, 0]
if args and args[0] == "--devices":
    devs = [int(x) for x in args[1].split(",")]
    args = args[2:]
sizes = [int(a) for a in args] or [12, 16]
ms = 3                                                         # 8 worker rows per challenge
for lg in sizes:
    rows = [codec.be32_to_fr_list(uniform_fr(1 << lg, 30 + k)) for k in range(1 << ms)]
    x = codec.be32_to_fr(uniform_fr(1, 2))
    out = {"log2_T": lg, "rows_per_challenge": 1 << ms}
    want = None
    for G in range(1, len(devs) + 1):
        m = MultiDeviceClient(devices=devs[:G], seed=5)
        m.start(scale=lg + ms, machines_scale=ms)
        got = [r.json() for r in m.commit_and_open_rows(range(1 << ms), rows, x)]      # warm-up + answer check
        assert all("proof" in g for g in got)
        assert want is None or got == want
        want = got
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:
            m.commit_and_open_rows(range(1 << ms), rows, x)
        reps = 20 if lg <= 16 else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            m.commit_and_open_rows(range(1 << ms), rows, x)
        dt = time.perf_counter() - t0
        out[f"contexts_{G}"] = {"devices": devs[:G], "requests_per_s": round(reps * (1 << ms) / dt, 1),
                                "ms_per_challenge": round(dt / reps * 1e3, 3)}
        m.stop()
    print(json.dumps(out), flush=True)
