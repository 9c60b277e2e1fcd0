#!/usr/bin/env python3
"""Per-kernel sums of arbitrary rocprofv3 --pmc counters: pmc_generic.py <counter_collection.csv>..."""
import collections
import csv
import sys

res = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        res[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
    for k, _ in seen:
        calls[(f, k)] += 1
for k in sorted(res):
    print(k)
    for c, v in sorted(res[k].items()):
        print(f"    {c:40s} {v:.6g}")
