#!/usr/bin/env python3
"""Window-size sweep for the KZG commit+open latency at small/medium row sizes (development aid; feeds choose_window)."""
import json
import subprocess
import sys

for lg in (int(a) for a in sys.argv[1:] or ["16"]):
    for c in range(max(6, lg - 6), min(22, lg + 3) + 1):
        out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--workload", "kzg22", "--no-cpu-baseline", "--log-n",
                              str(lg), "--window", str(c)], capture_output=True, text=True).stdout.strip().splitlines()
        try:
            d = json.loads(out[-1])
            st = d["stages_ms"]
            print(f"lg={lg} c={d['config']['window_bits']:2d} nwin={d['config']['windows']:2d} ms={d['ms_per_step']:.3f} "
                  f"sort={st['digits']:.2f} acc={st['accumulate']:.2f} fold={st['fixup']:.2f} tree={st['tree']:.2f} "
                  f"final={st['final']:.2f} ntt+poly={st['ntt'] + st['poly']:.2f}", flush=True)
        except Exception as e:  # noqa: BLE001
            print("lg", lg, "c", c, "failed", e, out[-1:] if out else "")
