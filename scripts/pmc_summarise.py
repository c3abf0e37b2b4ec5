"""Average the rocprofv3 --pmc counter values per kernel and dispatch (input: the directory scripts/pmc_round.sh
fills) into one CSV, and derive profiles/pmc_traffic.json for the dominant kernel (2 x FETCH_SIZE + WRITE_SIZE,
the gfx950 correction of MI355X_MICROARCH.md's HBM section)."""
import csv
import glob
import json
import os
import sys

accept = "--accept" in sys.argv      # a deliberate change of the kernel's traffic: take the new value without the drift gate
argv = [a for a in sys.argv[1:] if a != "--accept"]
src, out_csv = argv[0], argv[1]
acc = {}
for path in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        d = acc.setdefault(k, {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
with open(out_csv, "w") as f:
    f.write("# rocprofv3 --pmc <group> --kernel-trace, one pass per counter group (scripts/pmc_round.sh): python3 bench.py "
            "--steps 3 --warmup 1 --headline-only (2^20 MSM, c=20)\n")
    f.write("# raw counter values per dispatch, averaged over dispatches; FETCH_SIZE/WRITE_SIZE in KB; gfx950: FETCH_SIZE "
            "under-reports wide coalesced reads by 2x\n")
    f.write("kernel,counter,dispatches,avg_per_dispatch\n")
    for (kern, ctr), d in sorted(acc.items()):
        f.write(f"{kern},{ctr},{len(d)},{sum(d.values()) / len(d):.1f}\n")
fs = acc.get(("k_msm_accumulate", "FETCH_SIZE"))
ws = acc.get(("k_msm_accumulate", "WRITE_SIZE"))
if fs and ws:
    fkb, wkb = sum(fs.values()) / len(fs), sum(ws.values()) / len(ws)
    cfg = json.loads(open(glob.glob(os.path.join(src, "FETCH_SIZE.json"))[0]).read().strip().splitlines()[-1])["config"]
    iv = acc.get(("k_msm_accumulate", "SQ_INSTS_VALU"))
    # v_mad_u64_u32 wave-instructions per launch: ISA count per mixed addition (scripts/count_mads.py) x additions per
    # lane x waves (every lane sums `entries_per_lane` sorted entries)
    try:
        isa = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                          "isa_counts.json")))
        wave_mads = isa["mads_per_mixed_add"] * cfg["entries_per_lane"] * cfg["lanes"] / 64.0
    except (OSError, KeyError):
        wave_mads = None
    # staleness gate: bench.py quotes roofline.traffic from the COMMITTED file; a fresh pass that disagrees with it by more
    # than 2 % means the committed figure no longer describes the kernel (exit 3, file still rewritten from this pass)
    traffic_path = os.path.join(os.path.dirname(out_csv), "pmc_traffic.json")
    drift = None
    try:
        old_rec = json.load(open(traffic_path))
        if (old_rec.get("points_per_gpu"), old_rec.get("window_bits")) == (cfg["points_per_gpu"], cfg["window_bits"]):
            drift = (2 * fkb * 1024 + wkb * 1024) / old_rec["traffic_bytes_per_launch"] - 1.0
    except (OSError, KeyError, ValueError):
        pass
    json.dump({"workload": "msm20", "sq_insts_valu_per_launch": (sum(iv.values()) / len(iv)) if iv else None,
               "wave_mads_per_launch": wave_mads, "points_per_gpu": cfg["points_per_gpu"], "window_bits": cfg["window_bits"],
               "kernel": "k_msm_accumulate", "fetch_size_kb_raw": fkb, "write_size_kb_raw": wkb,
               "traffic_bytes_per_launch": 2 * fkb * 1024 + wkb * 1024,
               "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B; "
                             "MI355X_MICROARCH.md HBM section)",
               "drift_vs_previous_committed_value": drift,
               "source": os.path.relpath(out_csv)},
              open(traffic_path, "w"), indent=1)
    print("traffic bytes/launch", 2 * fkb * 1024 + wkb * 1024, "drift vs committed",
          "n/a" if drift is None else f"{drift * 100:+.2f} %")
    if drift is not None and abs(drift) > 0.02 and not accept:
        print("pmc_traffic.json was STALE: traffic per launch moved by more than 2 % (rerun with --accept if the kernel changed)")
        sys.exit(3)
