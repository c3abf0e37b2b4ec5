"""A/B of the batched-affine prototype (csrc/baff_proto.hip) against k_msm_accumulate on the same sorted entries
(dev tool; needs the GPU and `KZG_WITH_PROTO=1 python -m zkp_subnet_amd.build`, then KZG_MI355X_LIB=zkp_subnet_amd/libkzg_mi355x_proto.so).  Prints one JSON line per (size, lanes).   python scripts/proto_baff.py [log2_n ...]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import TAU, uniform_fr                              # noqa: E402
from zkp_subnet_amd import HipEngine                           # noqa: E402

_U32, _F = ctypes.c_uint32, ctypes.c_float
for lg in [int(a) for a in sys.argv[1:]] or [20, 22]:
    n = 1 << lg
    eng = HipEngine(0)
    eng._lib.kzg_proto_baff.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, _U32,
                                        ctypes.POINTER(_F), ctypes.POINTER(_U32)]
    eng.gen_srs(TAU, 1, lg, 0)
    eng.upload_fr(0, uniform_fr(n, 0), False)
    for lanes in (65536, 131072, 262144):
        ms = (ctypes.c_float * 8)()
        cnt = (ctypes.c_uint32 * 8)()
        best = None
        for _ in range(3):
            eng._chk(eng._lib.kzg_proto_baff(eng._h, 0, n, 0, lanes, ms, cnt))
            cur = list(ms)
            best = cur if best is None else [min(a, b) for a, b in zip(best, cur)]
        entries, np1, np2, np3 = cnt[0], cnt[1], cnt[2], cnt[3]
        acc_per_add_ns = best[1] * 1e6 / entries
        rec = {"log2_n": lg, "window_bits": eng.window, "lanes": lanes, "entries": entries,
               "accumulate_ms": round(best[1], 4), "accumulate_ns_per_add": round(acc_per_add_ns, 4),
               "baff_round_ms": [round(x, 4) for x in best[2:5]],
               "baff_ns_per_add": [round(best[2 + i] * 1e6 / p, 4) for i, p in enumerate((np1, np2, np3))],
               "pairs_per_lane": [round(p / lanes, 1) for p in (np1, np2, np3)],
               "mismatches_sampled": [cnt[4], cnt[5], cnt[6]], "equal_x_pairs_skipped": cnt[7],
               # 3 rounds do 7/8 of the additions; the last 1/8 (and the carries) would stay on the XYZZ kernel
               "projected_accumulate_ms": round(best[2] + best[3] + best[4] + best[1] / 8, 4)}
        print(json.dumps(rec), flush=True)
    eng.close()
