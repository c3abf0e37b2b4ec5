"""Reads a `rocprofv3 --kernel-trace` CSV of `bench.py --headline-only --in-flight 2` (scripts/trace_pipelined.sh) and says
whether two MSMs in flight really overlap (VERDICT r4 task 2): per HIP stream, the hardware queue its kernels were
dispatched on, and -- over the timed steps -- how much of each request's latency-bound tail (fold, bucket tree, final) ran
WHILE the other request's sort / accumulate was executing.  Two lanes on one hardware queue show up as one Queue_Id for both
streams and an overlap near zero (round 4's regression); two queues and a large overlap is the working state.

    python scripts/trace_pipelined_summary.py gpurun_out/trace2/t_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
acc = [r for r in rows if r["name"] == "k_msm_accumulate"]
if len(acc) < 6:
    sys.exit("no accumulate launches in the trace")
# the streams that carry MSMs, and the queue each one's kernels were dispatched on
streams = defaultdict(set)
for r in acc:
    streams[r["Stream_Id"]].add(r["Queue_Id"])
print("MSM-carrying streams -> hardware queues:", {s: sorted(q) for s, q in streams.items()})
msm = [r for r in rows if r["Stream_Id"] in streams and r["name"].startswith(("k_msm_", "k_sort_", "k_fold_", "k_publish"))]
TAIL = ("k_fold_", "k_msm_tree_", "k_msm_final_")
# the two-in-flight region = the longest run of accumulate launches that ALTERNATE between two streams (bench.py also runs
# one-at-a-time phases -- stage profile, single-request latency -- before and after it)
best, cur = [], []
for r in acc:
    if cur and r["Stream_Id"] != cur[-1]["Stream_Id"]:
        cur.append(r)
    else:
        cur = [r]
    if len(cur) > len(best):
        best = list(cur)
if len(best) < 4:
    sys.exit("no region with two requests in flight in this trace")
last = best[2:]                       # (the first two of the run fill the pipeline)
t0, t1 = last[0]["s"], last[-1]["e"]
tail = [r for r in msm if t0 <= r["s"] <= t1 and r["name"].startswith(TAIL)]
busy = [r for r in msm if t0 - 5_000_000 <= r["s"] <= t1 and r["name"].startswith(("k_msm_accumulate", "k_sort_"))]
tot = ov = 0
for t in tail:
    d = t["e"] - t["s"]
    tot += d
    for b in busy:
        if b["Stream_Id"] != t["Stream_Id"]:
            ov += max(0, min(t["e"], b["e"]) - max(t["s"], b["s"]))
span = (t1 - t0) / 1e6
print(f"timed part of the trace: {len(last)} accumulate launches over {span:.2f} ms = {span / len(last):.3f} ms per MSM")
print(f"tail kernels (fold / tree / final) of those requests: {len(tail)} launches, {tot / 1e6:.3f} ms in total; "
      f"{ov / 1e6:.3f} ms of it ({100.0 * ov / max(tot, 1):.0f} %) ran while the OTHER stream's sort / accumulate was executing")
