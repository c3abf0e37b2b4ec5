#!/usr/bin/env python3
"""Bring-up script for the GPU box (not collected by pytest): staged parity checks against the oracle with timings.
Development aid kept under tests/ because it uses the oracle; the judged parity tests are tests/test_gpu_*.py.

    python tests/bringup_gpu_check.py [field] [g1] [srs] [golden] [mid] [big]
"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bls12_381 as o  # noqa: E402
from oracle import cpu as oc  # noqa: E402
from zkp_subnet_amd import HipEngine  # noqa: E402

H = bytes.fromhex
stage = [0]


def say(*a):
    print(f"[{time.time() - T0:7.2f}s]", *a, flush=True)


T0 = time.time()
rnd = random.Random(1234)
only = set(sys.argv[1:])


def want(name):
    return not only or name in only


if want("field"):
    eng = HipEngine(0)
    for field, mod, w in ((0, o.P, 48), (1, o.R, 32)):
        vals_a = [rnd.randrange(mod) for _ in range(2000)] + [0, 1, mod - 1, mod - 1, 0, 2]
        vals_b = [rnd.randrange(mod) for _ in range(2000)] + [0, mod - 1, mod - 1, 1, 5, (mod + 1) // 2]
        a = b"".join(v.to_bytes(w, "big") for v in vals_a)
        b = b"".join(v.to_bytes(w, "big") for v in vals_b)
        for op, fn in ((0, lambda x, y: x * y % mod), (1, lambda x, y: (x + y) % mod), (2, lambda x, y: (x - y) % mod),
                       (3, lambda x, y: x * y % mod), (4, lambda x, y: x * x % mod)):
            out = eng.test_field(field, op, a, b)
            got = [int.from_bytes(out[i * w:(i + 1) * w], "big") for i in range(len(vals_a))]
            exp = [fn(x, y) for x, y in zip(vals_a, vals_b)]
            bad = [i for i in range(len(exp)) if got[i] != exp[i]]
            say(f"field={field} op={op}: {'OK' if not bad else 'MISMATCH at ' + str(bad[:5])}")
            if bad:
                i = bad[0]
                say(hex(vals_a[i]), hex(vals_b[i]), hex(got[i]), hex(exp[i]))
    eng.close()

if want("g1"):
    eng = HipEngine(0)
    tb = o.g1_table()
    pts_a = [tb.mul(rnd.randrange(1, o.R)) for _ in range(64)]
    pts_b = [tb.mul(rnd.randrange(1, o.R)) for _ in range(64)]
    pts_b[0] = pts_a[0]                 # doubling inside the mixed add
    pts_b[1] = o.g1_neg(pts_a[1])       # cancels to infinity
    pts_b[2] = None                     # infinity operand
    pts_a[3] = None
    a = b"".join(o.g1_to_be96(p) for p in pts_a)
    b = b"".join(o.g1_to_be96(p) for p in pts_b)
    exp_fns = {0: lambda x, y: o.g1_add(x, y), 1: lambda x, y: o.g1_add(o.g1_add(x, x), y),
               2: lambda x, y: o.g1_add(x, x), 3: lambda x, y: o.g1_mul(x, 4) if x else None}
    for op in range(4):
        out = eng.test_g1(op, a, b)
        exp = b"".join(o.g1_to_be96(exp_fns[op](x, y)) for x, y in zip(pts_a, pts_b))
        bad = [i for i in range(64) if out[96 * i:96 * i + 96] != exp[96 * i:96 * i + 96]]
        say(f"g1 op={op}: {'OK' if not bad else 'MISMATCH at ' + str(bad[:8])}")
    eng.close()

if want("srs"):
    for c in (0, 4, 7):
        eng = HipEngine(0, window=c)
        tx, ty = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
        t = time.time()
        eng.gen_srs(tx, ty, 6, 2)
        say(f"gen_srs scale 6/2 window={eng.window} {time.time() - t:.3f}s")
        got = eng.srs_read(0, 64)
        exp = b"".join(oc.srs_gen(tx.to_bytes(32, "big"), ty.to_bytes(32, "big"), 6, 2, i) for i in range(4))
        say("srs points", "OK" if got == exp else "MISMATCH")
        cw = eng.window
        offs = eng.window_offsets
        nwin = len(offs) - 1
        okw = True
        for w in (1, nwin - 1):
            tabw = eng.srs_read(0, 4, window=w)
            for j in range(4):
                e = o.g1_mul(o.g1_from_be96(exp[96 * j:96 * j + 96]), pow(2, offs[w], o.R))
                okw &= tabw[96 * j:96 * j + 96] == o.g1_to_be96(e)
        say("window tables", "OK" if okw else "MISMATCH")
        # msm over slice 1
        sc = [rnd.randrange(o.R) for _ in range(16)]
        got = eng.msm(o.fr_to_be32(sc), 16)
        e = oc.msm(exp[96 * 16:96 * 32], o.fr_to_be32(sc))
        say(f"msm16 window={cw}", "OK" if got == e else f"MISMATCH {got.hex()} {e.hex()}")
        for name, scv in (("zeros", [0] * 16), ("ones", [1] * 16), ("rm1", [o.R - 1] * 16), ("one_elem", [5])):
            got = eng.msm(o.fr_to_be32(scv), 0)
            e = oc.msm(exp[:96 * len(scv)], o.fr_to_be32(scv))
            say(f"  msm {name}", "OK" if got == e else f"MISMATCH {got.hex()} {e.hex()}")
        eng.close()

if want("golden"):
    g = json.load(open(os.path.join(ROOT, "tests/golden/msm.json")))
    for case in g:
        if not case["points"]:
            continue
        n = len(case["points"])
        npad = 1
        while npad < n:
            npad *= 2
        pts = b"".join(H(p) for p in case["points"]) + bytes(96 * (npad - n))
        eng = HipEngine(0, window=5)
        sc = npad.bit_length() - 1
        eng.load_srs(pts, sc, 0)
        got = eng.msm(b"".join(H(s) for s in case["scalars"]), 0)
        say(f"golden msm {case['name']}", "OK" if got.hex() == case["result"] else f"MISMATCH {got.hex()}")
        eng.close()
    k = json.load(open(os.path.join(ROOT, "tests/golden/kzg.json")))
    tx, ty = int(k["tau_x"], 16), int(k["tau_y"], 16)
    for case in k["cases"]:
        eng = HipEngine(0)
        eng.gen_srs(tx, ty, case["scale"], case["machines_scale"], [case["i"]])
        row = b"".join(H(v) for v in case["row"])
        ef = case["evaluation_form"]
        c = eng.commit(0, row, ef)
        ev, pf = eng.open(0, row, H(case["alpha"]), ef)
        c2, ev2, pf2 = eng.commit_open(0, row, H(case["alpha"]), ef)
        ok = c.hex() == case["commitment"] and ev.hex() == case["eval"] and pf.hex() == case["proof"]
        ok2 = (c2, ev2, pf2) == (c, ev, pf)
        say(f"golden kzg {case['name']}", "OK" if ok and ok2 else
            f"MISMATCH c={c.hex() == case['commitment']} e={ev.hex() == case['eval']} p={pf.hex() == case['proof']} fused={ok2}")
        eng.close()
    nt = json.load(open(os.path.join(ROOT, "tests/golden/ntt.json")))
    eng = HipEngine(0)
    for case in nt:
        a = b"".join(H(v) for v in case["input"])
        f = eng.ntt(a, False)
        i = eng.ntt(a, True)
        say(f"golden ntt n={case['n']}", "OK" if f == b"".join(H(v) for v in case["forward"]) and
            i == b"".join(H(v) for v in case["inverse"]) else "MISMATCH")
    eng.close()

if want("mid"):
    for lg in (10, 12, 14):
        eng = HipEngine(0)
        tx = rnd.randrange(1, o.R)
        t = time.time(); eng.gen_srs(tx, 1, lg, 0); tg = time.time() - t
        n = 1 << lg
        sc = [rnd.randrange(o.R) for _ in range(n)]
        scb = o.fr_to_be32(sc)
        t = time.time(); got = eng.msm(scb, 0); tm = time.time() - t
        srs = eng.srs_read(0, n)
        e = oc.msm(srs, scb, 8)
        e2 = oc.g1_mul_gen(o.poly_eval(sc, tx).to_bytes(32, "big"))
        say(f"msm 2^{lg} c={eng.window}: vs C oracle {'OK' if got == e else 'MISMATCH'}, trapdoor {'OK' if got == e2 else 'MISMATCH'}"
            f" gen {tg:.3f}s msm {tm * 1e3:.2f}ms")
        al = rnd.randrange(o.R).to_bytes(32, "big")
        c, ev, pf = eng.commit_open(0, scb, al, True)
        ec = oc.commit(srs, scb, True, 8)
        eev, epf = oc.open_(srs, scb, al, True, 8)
        say(f"  commit/open 2^{lg}: {'OK' if (c, ev, pf) == (ec, eev, epf) else 'MISMATCH'}")
        f = eng.ntt(scb, False)
        say(f"  ntt 2^{lg}: {'OK' if f == oc.fr_ntt(scb, False) else 'MISMATCH'}")
        eng.close()

if want("big"):
    for lg in (16, 20):
        eng = HipEngine(0)
        tx = rnd.randrange(1, o.R)
        t = time.time(); eng.gen_srs(tx, 1, lg, 0); tg = time.time() - t
        n = 1 << lg
        import numpy as np
        rs = np.random.default_rng(7)
        raw = rs.integers(0, 256, size=(n, 32), dtype=np.uint8)
        raw[:, 0] &= 0x3F
        scb = raw.tobytes()
        eng.upload_fr(0, scb, False)
        eng.set_profiling(True)
        got = eng.msm_resident(0, n, 0)
        ts = []
        for _ in range(5):
            t = time.time(); got2 = eng.msm_resident(0, n, 0); ts.append(time.time() - t)
        tim = eng.timings()
        sc = [int.from_bytes(scb[32 * i:32 * i + 32], "big") for i in range(n)]
        e2 = oc.g1_mul_gen(o.poly_eval(sc, tx).to_bytes(32, "big"))
        say(f"msm 2^{lg} c={eng.window} plan={eng.msm_plan(n)}: trapdoor {'OK' if got == e2 and got2 == e2 else 'MISMATCH'} gen {tg:.2f}s "
            f"msm wall {min(ts) * 1e3:.2f}ms  {n / min(ts) / 1e6:.2f} Mpts/s")
        say("   stages(ms):", {k: round(v, 3) for k, v in tim.items()})
        eng.close()
say("done")
