"""One validator step at the reference's MAINNET scale (scale 24 / machines_scale 8: 256 rows of 2^16; reference
neurons/validator.py:58-133, Makefile:63-74) on this library, timed part by part (needs the GPU):
  random_poly()                 2^24 uniform field elements as wire text (native generator; the 30 s challenge deadline
                                of neurons/validator.py:206 is the yardstick -- a Python loop needs ~40 s for this alone)
  generate_challenge(client,256) random_poly + random_point + 256 x eval(fft(row, inverse), alpha)
  256 x Miner.forward           the miner side of the same step, one GPU, the UNCHANGED two-call route
  verify_all                    all 256 rows in ONE batched pairing check (random linear combination; per-row work on a host
                                thread pool); beside it the row-by-row form (256 pairing checks on the same pool)
    python scripts/validator_step.py [--scale 24] [--machines-scale 8] [--out profiles/r03_validator_step.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkp_subnet_amd.client import Client                      # noqa: E402
from zkp_subnet_amd.miner import Miner, default_config        # noqa: E402
from zkp_subnet_amd.validator import generate_challenge, reward, verify_all   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--machines-scale", type=int, default=8)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--out", default="")
a = ap.parse_args()
rows = 1 << a.machines_scale
cl = Client(seed=11)
t0 = time.perf_counter()
cl.start(a.scale, a.machines_scale)
start_s = time.perf_counter() - t0
res = {"scale": a.scale, "machines_scale": a.machines_scale, "rows": rows, "row_length": 1 << (a.scale - a.machines_scale),
       "host_threads": a.threads, "client_start_synthetic_s": round(start_s, 3)}
t0 = time.perf_counter()
with cl.random_poly() as r:
    poly = r.json()["poly"]
res["random_poly_s"] = round(time.perf_counter() - t0, 3)
assert len(poly) == rows
del poly
t0 = time.perf_counter()
ch = generate_challenge(cl, rows)
res["generate_challenge_s"] = round(time.perf_counter() - t0, 3)
# the challenge is 2^24 str objects in 256 lists: park it in the permanent generation, or every later full collection walks
# 16.7 M pointers (~0.1 s) in the middle of whichever timing happens to trigger it
import gc                                                     # noqa: E402

gc.collect()
gc.freeze()
miner = Miner(default_config(scale=a.scale, machines_scale=a.machines_scale), client=cl)
t0 = time.perf_counter()
responses = [miner.forward(ch.to_synapse(i)) for i in range(rows)]
res["miner_forward_all_rows_s"] = round(time.perf_counter() - t0, 3)
res["miner_forward_per_row_ms"] = round(res["miner_forward_all_rows_s"] / rows * 1e3, 3)
t0 = time.perf_counter()
ok = verify_all(cl, ch, responses, threads=a.threads)
res["verify_all_s"] = round(time.perf_counter() - t0, 3)
assert all(ok), "a proof failed to verify"
from concurrent.futures import ThreadPoolExecutor             # noqa: E402

t0 = time.perf_counter()
with ThreadPoolExecutor(max_workers=a.threads) as ex:
    rowwise = list(ex.map(lambda i: reward(cl, ch, responses[i], i, 0.0) == 1.0, range(rows)))
res["verify_row_by_row_on_pool_s"] = round(time.perf_counter() - t0, 3)
assert all(rowwise)
spoiled = list(responses)
spoiled[rows // 2] = spoiled[rows // 2].model_copy(update={"proof": responses[0].proof})
t0 = time.perf_counter()
ok2 = verify_all(cl, ch, spoiled, threads=a.threads)           # batch fails -> row by row finds the culprit
res["verify_all_with_one_bad_row_s"] = round(time.perf_counter() - t0, 3)
assert ok2 == [i != rows // 2 for i in range(rows)]
t0 = time.perf_counter()
one = [reward(cl, ch, responses[i], i, 0.0) for i in range(min(rows, 16))]
res["worker_verify_serial_ms_per_row"] = round((time.perf_counter() - t0) / len(one) * 1e3, 3)
assert all(v == 1.0 for v in one)
res["step_total_s"] = round(res["generate_challenge_s"] + res["miner_forward_all_rows_s"] + res["verify_all_s"], 3)
print(json.dumps(res), flush=True)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
cl.stop()
