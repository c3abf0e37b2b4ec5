"""What GPU_MAX_HW_QUEUES costs when TWO processes share one GPU (a miner and a validator on a development box; the
one-GPU self-tests of the multi-rank bench): each of two processes runs 2^20 MSMs back to back for ~3 s; prints the MSM rate of
each.  Usage: python scripts/ab_hw_queues_two_processes.py [queues ...]   (GPU box)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    from bench import TAU, uniform_fr
    from zkp_subnet_amd import HipEngine

    n = 1 << 20
    eng = HipEngine(0)
    eng.gen_srs(TAU, 1, 20, 0)
    eng.upload_fr(0, uniform_fr(n, seed=0), False)
    for _ in range(10):
        eng.msm_resident(0, n, 0)
    # start together: wait for the wall clock to reach the agreed second
    while time.time() < float(sys.argv[2]):
        time.sleep(0.001)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 3.0:
        eng.msm_resident(0, n, 0)
        k += 1
    print(json.dumps({"ms_per_msm": (time.perf_counter() - t0) / k * 1e3}), flush=True)
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    for q in (sys.argv[1:] or ["4", "6", "8"]):
        for procs in (1, 2):
            env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
            start = str(time.time() + 12.0)
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", start], env=env, stdout=subprocess.PIPE,
                                   stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
            outs = [json.loads(p.communicate()[0].strip().splitlines()[-1])["ms_per_msm"] for p in ps]
            print(f"GPU_MAX_HW_QUEUES={q} processes={procs}: ms per 2^20 MSM in each process: " + ", ".join(f"{o:.2f}" for o in outs), flush=True)
