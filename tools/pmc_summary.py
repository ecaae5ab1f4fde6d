#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel name. Usage: tools/pmc_summary.py <dir> [name-substring]"""
import collections, csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if flt and flt not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    print(k, "calls", len(calls[k]))
    for c, x in sorted(v.items()):
        print(f"   {c:32s} {x:.4g}")
