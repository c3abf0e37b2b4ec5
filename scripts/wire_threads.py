import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zkp_subnet_amd import codec
w = codec._wire
for lg in (12, 14, 16, 18, 20):
    n = 1 << lg
    lst = w.encode_fr_list(os.urandom(32 * n))
    for th in (1, 2, 4, 8, 16):
        best = 1e9
        for _ in range(7):
            t = time.perf_counter(); w.decode_fr_list(lst, th); best = min(best, time.perf_counter() - t)
        print(lg, th, round(best * 1e3, 3), "ms")
