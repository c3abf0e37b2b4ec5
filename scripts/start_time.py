"""Times the reference's production start path at the reference's sizes: `Client(setup_path=...).start(scale,
machines_scale)` from a setup FILE (reference base/miner.py:75-84; mainnet = Makefile:63-74, setup_24_8: 2^24 points,
1.6 GB uncompressed; testnet = Makefile:89-101, 20 / 8).  The file is generated first (zkp_subnet_amd.setup_cli, timed
separately); every start is a fresh Client / context.  Needs the GPU.

    python scripts/start_time.py [--configs 24:8,20:8] [--dir /tmp] [--out profiles/r03_start_time.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkp_subnet_amd import setup_cli                      # noqa: E402
from zkp_subnet_amd.client import Client                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="24:8,20:8")
ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
ap.add_argument("--out", default="")
ap.add_argument("--repeat", type=int, default=2)
a = ap.parse_args()
res = {"what": "Client(setup_path).start(scale, machines_scale): mmap -> pinned double-buffered tiles -> H2D -> decode "
               "(compressed: one Fp square root per point) -> window tables; seconds", "rows": []}
for cfg in a.configs.split(","):
    scale, ms = (int(v) for v in cfg.split(":"))
    for compressed in (False, True):
        path = os.path.join(a.dir, f"setup_{scale}_{ms}." + ("compressed" if compressed else "uncompressed"))
        t0 = time.perf_counter()
        rc = setup_cli.main(["setup", "--setup-path", path, "--scale", str(scale), "--machines-scale", str(ms),
                             "--generate-setup", "--overwrite", "--seed", "7"] + (["--compressed"] if compressed else []))
        assert rc == 0
        gen_s = time.perf_counter() - t0
        runs = []
        for _ in range(a.repeat):
            cl = Client(setup_path=path, uncompressed=not compressed)
            t0 = time.perf_counter()
            cl.start(scale, ms)
            wall = time.perf_counter() - t0
            st = cl.engine.load_stats()
            runs.append({"start_s": round(wall, 3), **{k: round(v, 3) for k, v in st.items()}})
            window, pts = cl.engine.window, cl.engine.srs_points
            cl.stop()
        row = {"scale": scale, "machines_scale": ms, "points": pts, "file_bytes": os.path.getsize(path),
               "compressed": compressed, "window_bits": window, "setup_cli_generate_and_write_s": round(gen_s, 2),
               "starts": runs, "start_s": min(r["start_s"] for r in runs)}
        print(json.dumps(row), flush=True)
        res["rows"].append(row)
        os.remove(path)
        os.remove(path + ".vk")
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
