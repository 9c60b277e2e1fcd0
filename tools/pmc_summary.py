#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel.
usage: pmc_summary.py out.json fetch_counter_collection.csv write_counter_collection.csv
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch (rocprofv3 derived metric).  Per the MI355X guide,
on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x; the kernels here read
mostly 4-12 B per lane, which the guide calls uncalibrated -- raw numbers are reported and the x2
bound is given beside them."""
import collections
import csv
import json
import sys


def agg(path):
    out = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        out[(k, r["Counter_Name"])][0] += float(r["Counter_Value"])
        out[(k, r["Counter_Name"])][1] += 1
    return out


def main():
    dst, files = sys.argv[1], sys.argv[2:]
    res = collections.defaultdict(dict)
    for f in files:
        for (k, c), (v, n) in agg(f).items():
            res[k][c + "_KiB_total"] = v
            res[k][c + "_dispatches"] = n
            res[k][c + "_bytes_per_launch"] = v * 1024.0 / max(1, n)
            res[k][c + "_bytes_total"] = v * 1024.0
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each (tools/profile_bench.sh), of ONE bench.py step of this "
                    "workload; KiB per dispatch summed per kernel; raw counters (calibration: profiles/r04_pmc_calibration.txt -- "
                    "FETCH_SIZE counts 64 B per request, streamed reads move 128 B per request: true reads lie between x1 and x2)")
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items()):
        if not isinstance(v, dict):
            continue
        print(k, {a: round(b) for a, b in v.items() if a.endswith("per_launch")})


if __name__ == "__main__":
    main()
