#!/usr/bin/env python3
"""Reads a driver SCALE record (SCALE_rNN.json) -- or any file holding bench.py lines -- and fails when the multi-GPU numbers
leave the band DESIGN.md section 4 states for them (VERDICT r5 task 1).  The record's exact layout is the driver's: this script
walks the JSON, takes every object that looks like a bench.py line ("n_gpus" + "ms_per_step"), and also re-parses full lines
out of any captured stdout ("tail" strings), because the driver's `parsed` form keeps only the contract keys.

Checks (each prints PASS / FAIL / SKIP with the number it looked at):
  1. no line's `config.collective` starts with "gloo fallback" (the host-side exchange was measured, not RCCL)
  2. `rccl_version` is a real version at N > 1
  3. weak scaling (msm20, 2^20 points per GPU): ms_per_step at every N <= 3.0 ms and <= 1.10 x the N = 1 step
  4. msm26 (2^26 points over N segments, strong scaling): efficiency at N = 8 >= 85 %, at N = 4 >= 90 %, at N = 2 >= 93 %
     (t_1 = the record's own N = 1 msm26 step if present, else 124.7 ms x (this record's N = 1 msm20 step / 2.79 ms): the
     one-box size sweep DESIGN quotes, rescaled to the record's box)
  5. pianist_kzg22 (one 2^22 commit+open per GPU, nothing exchanged): per-segment latency at N <= 1.10 x the smallest seen

Exit code 0: every applicable check passed (or the record says "skipped": nothing to judge); 1: a check failed; 2: unreadable.

    python scripts/check_scale.py SCALE_r06.json
"""
import json
import sys

WEAK_STEP_MAX_MS = 3.0
WEAK_STEP_GROWTH = 1.10
MSM26_EFF_MIN = {2: 0.93, 4: 0.90, 8: 0.85}
MSM26_ONE_GPU_MS, MSM26_BOX_MSM20_MS = 124.7, 2.79          # profiles/r04_sweep_msm_2_20_to_2_26.log
PIANIST_GROWTH = 1.10


def lines_in(obj, out):
    """every dict that looks like a bench.py line, anywhere in the record; full lines hidden in captured stdout too"""
    if isinstance(obj, dict):
        if "n_gpus" in obj and "ms_per_step" in obj and "metric" in obj:
            out.append(obj)
        for v in obj.values():
            lines_in(v, out)
    elif isinstance(obj, list):
        for v in obj:
            lines_in(v, out)
    elif isinstance(obj, str) and '"n_gpus"' in obj:
        for ln in obj.splitlines():
            ln = ln.strip()
            if ln.startswith("{") and ln.endswith("}"):
                try:
                    lines_in(json.loads(ln), out)
                except ValueError:
                    pass


def main(path):
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError) as e:
        print(f"cannot read {path}: {e}")
        return 2
    if isinstance(rec, dict) and rec.get("skipped"):
        print(f"SKIP  the record was skipped by the driver ({rec.get('reason', 'no reason given')}): nothing to judge")
        return 0
    found = []
    lines_in(rec, found)
    # headline lines (msm20 per GPU) keyed by N; the richest line per N wins (the full stdout line carries msm26 / pianist)
    head = {}
    for ln in found:
        if "points/sec" not in str(ln.get("metric", "")) or "msm26" in str(ln.get("config", {}).get("workload", "")):
            continue
        n = int(ln["n_gpus"])
        if n not in head or len(ln) > len(head[n]):
            head[n] = ln
    if not head:
        print("FAIL  no bench.py headline line found in the record")
        return 1
    ok = True

    def verdict(passed, text):
        nonlocal ok
        ok = ok and passed
        print(("PASS  " if passed else "FAIL  ") + text)

    for n in sorted(head):
        cfg = head[n].get("config", {})
        coll = str(cfg.get("collective") or "")
        if n > 1:
            verdict(not coll.startswith("gloo fallback") and "fallback" not in coll.split("(")[0],
                    f"N={n}: collective = {coll[:110] or 'none recorded'}")
            rv = str(cfg.get("rccl_version") or "")
            verdict(bool(rv) and not rv.startswith("none"), f"N={n}: rccl_version = {rv[:60] or 'missing'}")
    t1 = head.get(1, {}).get("ms_per_step")
    for n in sorted(head):
        t = float(head[n]["ms_per_step"])
        lim = WEAK_STEP_MAX_MS if not t1 else min(WEAK_STEP_MAX_MS, max(WEAK_STEP_GROWTH * float(t1), 0.0))
        verdict(t <= lim or n == 1 and t <= WEAK_STEP_MAX_MS,
                f"N={n}: weak-scaling step {t:.3f} ms (limit {lim:.3f}); value {head[n].get('value', 0) / 1e6:.1f} M points/s")
    msm26 = {n: head[n]["msm26"] for n in head if isinstance(head[n].get("msm26"), dict)}
    if not msm26:
        print("SKIP  no msm26 object in the record (the driver kept only the contract keys, or the launch had no process group)")
    t26_1 = None
    for n in sorted(msm26):
        m = msm26[n]
        if "error" in m:
            verdict(False, f"N={n}: msm26 reported {m['error']}")
            continue
        t = float(m["ms_per_step"])
        if int(m.get("n_gpus", n)) == 1:
            t26_1 = t
    if t26_1 is None and t1:
        t26_1 = MSM26_ONE_GPU_MS * float(t1) / MSM26_BOX_MSM20_MS
    for n in sorted(msm26):
        m = msm26[n]
        if "error" in m or n == 1 or not t26_1:
            continue
        eff = t26_1 / (n * float(m["ms_per_step"]))
        want = MSM26_EFF_MIN.get(n, 0.85)
        verdict(eff >= want, f"N={n}: msm26 {float(m['ms_per_step']):.2f} ms per step, strong-scaling efficiency {eff * 100:.1f} % "
                              f"(>= {want * 100:.0f} % wanted; t_1 = {t26_1:.1f} ms)")
    pian = {n: head[n]["pianist_kzg22"] for n in head if isinstance(head[n].get("pianist_kzg22"), dict)}
    good = {n: float(p["ms_per_step"]) for n, p in pian.items() if "error" not in p}
    for n, p in sorted(pian.items()):
        if "error" in p:
            verdict(False, f"N={n}: pianist_kzg22 reported {p['error']}")
    if good:
        base = min(good.values())
        for n in sorted(good):
            verdict(good[n] <= PIANIST_GROWTH * base, f"N={n}: pianist_kzg22 {good[n]:.2f} ms per segment (best {base:.2f})")
    else:
        print("SKIP  no pianist_kzg22 object in the record")
    # the curve as one table (the driver computes efficiency itself; this is for the reader of the log)
    print("N  msm20 ms/step  M points/s  weak eff.   msm26 ms  strong eff.   pianist ms")
    v1 = float(head[1]["value"]) if 1 in head else None
    for n in sorted(head):
        h = head[n]
        weak = f"{float(h['value']) / (n * v1) * 100:6.1f} %" if v1 else "     - "
        m = msm26.get(n, {})
        m_ms = f"{float(m['ms_per_step']):8.2f}" if "ms_per_step" in m else "       -"
        m_eff = f"{t26_1 / (n * float(m['ms_per_step'])) * 100:6.1f} %" if "ms_per_step" in m and t26_1 else "     - "
        pk = f"{good[n]:8.2f}" if n in good else "       -"
        print(f"{n:<2} {float(h['ms_per_step']):12.3f} {float(h['value']) / 1e6:11.1f}  {weak}   {m_ms}   {m_eff}    {pk}")
    print("RESULT " + ("inside DESIGN section 4's band" if ok else "OUTSIDE the band: look at the collective step first (DESIGN section 4)"))
    return 0 if ok else 1


if __name__ == "__main__":
    if len(sys.argv) != 2:
        print(__doc__)
        sys.exit(2)
    sys.exit(main(sys.argv[1]))
