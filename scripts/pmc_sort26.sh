#!/bin/bash
# HBM traffic of the sort kernels at 2^24 .. 2^26 (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass each).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/pmc26
ARGS="bench.py --workload msm26 --log-n ${LOGN:-26} --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined"
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc26/$grp -o p --output-format csv -- python3 $ARGS > gpurun_out/pmc26/$grp.json 2> gpurun_out/pmc26/$grp.err
done
python3 - <<'PY'
import csv, glob
acc = {}
for path in glob.glob("gpurun_out/pmc26/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        d = acc.setdefault(k, {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for (kern, ctr), d in sorted(acc.items()):
    if "sort" in kern or "accumulate" in kern:
        print(f"{kern:28s} {ctr:12s} dispatches={len(d):3d} avg_GB={sum(d.values()) / len(d) / 1e6:9.3f}")
PY
