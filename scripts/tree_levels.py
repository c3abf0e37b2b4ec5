"""Median duration of every bucket-tree / fold / final launch of a rocprofv3 --kernel-trace (csv) directory, by kernel and grid."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if "tree_level" in n or "fold" in n or "final" in n:
        d[(n, int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    print("%-28s grid %8d  n %3d  median %7.1f us" % (k[0], k[1], len(v), sorted(v)[len(v) // 2] / 1e3))
