"""What ONE device of a G-GPU host pays at start when it loads only the slices of its own worker indices
(`kzg_load_srs_file_slices`, VERDICT r5 task 4) against the whole setup file: the reference's mainnet file (scale 24 /
machines_scale 8: 2^24 points, 1.6 GB; Makefile:63-74), device g = 0 of G = 1 (the whole file), 2, 4, 8 -- each a fresh
context on this box's one GPU (the other G - 1 devices of a real host do the same work in parallel on their own GPUs and their
own share of the page cache).  Also the whole `MultiDeviceClient.start` over G contexts of this one GPU (serialised by the GPU:
an upper bound for a real host).  Needs the GPU.

    python scripts/start_time_slices.py [--config 24:8] [--dir /tmp] [--out profiles/r06_start_time_slices.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkp_subnet_amd import MultiDeviceClient, setup_cli   # noqa: E402
from zkp_subnet_amd.client import Client                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="24:8")
ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
ap.add_argument("--out", default="")
a = ap.parse_args()
scale, ms = (int(v) for v in a.config.split(":"))
path = os.path.join(a.dir, f"setup_{scale}_{ms}.uncompressed")
assert setup_cli.main(["setup", "--setup-path", path, "--scale", str(scale), "--machines-scale", str(ms), "--generate-setup",
                       "--overwrite", "--seed", "7"]) == 0
M = 1 << ms
res = {"what": "seconds of Client.start from the setup file when the context loads only the slices i = 0 (mod G) of it",
       "scale": scale, "machines_scale": ms, "file_bytes": os.path.getsize(path), "rows": []}
for G in (1, 2, 4, 8):
    runs = []
    for _ in range(2):
        cl = Client(setup_path=path, workers=list(range(0, M, G)) if G > 1 else None)
        t0 = time.perf_counter()
        cl.start(scale, ms)
        wall = time.perf_counter() - t0
        runs.append({"start_s": round(wall, 3), **{k: round(v, 3) for k, v in cl.engine.load_stats().items()}})
        pts, nwin = cl.engine.srs_points, len(cl.engine.window_offsets) - 1
        cl.stop()
    row = {"G": G, "resident_points": pts, "windows": nwin, "table_gb": round(pts * 128 * nwin / 1e9, 2),
           "starts": runs, "start_s": min(r["start_s"] for r in runs)}
    print(json.dumps(row), flush=True)
    res["rows"].append(row)
for G in (2, 4):
    mc = MultiDeviceClient([0] * G, setup_path=path)
    t0 = time.perf_counter()
    mc.start(scale, ms)
    wall = time.perf_counter() - t0
    pts = [c.engine.srs_points for c in mc.clients]
    mc.stop()
    row = {"multi_device_client_contexts_on_one_gpu": G, "start_s": round(wall, 3), "resident_points_per_context": pts}
    print(json.dumps(row), flush=True)
    res["rows"].append(row)
os.remove(path)
os.remove(path + ".vk")
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
