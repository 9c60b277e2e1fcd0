#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts (tools/run_bench.hip `calib`).
usage: pmc_calibrate.py calib_stdout.txt fetch_counter_collection.csv write_counter_collection.csv
Every k_calib dispatch touches `bytes` useful bytes exactly once in runs of `run` bytes (run-aligned, pseudo-random
places of a 16-GiB buffer), `word` bytes per lane.  Prints counter / bytes per pattern (the counters are in KiB).  Modes with a * are runs that start at any multiple of 8 bytes
(k_runs), the others start at multiples of their own length."""
import csv
import sys


def counters(path, name):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name and ("k_calib" in r["Kernel_Name"] or "k_runs" in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) * 1024.0 for r in rows]


def main():
    known = []
    for line in open(sys.argv[1]):
        if line.startswith("CALIB"):
            p = line.split()
            known.append((p[2] + ("*" if len(p) > 6 else ""), int(p[3].split("=")[1]), int(p[4].split("=")[1]), float(p[5].split("=")[1])))
    fetch = counters(sys.argv[2], "FETCH_SIZE")
    write = counters(sys.argv[3], "WRITE_SIZE")
    assert len(fetch) == len(known) == len(write), (len(fetch), len(write), len(known))
    print(f"{'mode':6s} {'B/lane':>6s} {'run B':>7s} {'useful GB':>10s} {'FETCH/useful':>13s} {'WRITE/useful':>13s}")
    for (mode, word, run, b), f, w in zip(known, fetch, write):
        print(f"{mode:6s} {word:6d} {run:7d} {b / 1e9:10.2f} {f / b:13.3f} {w / b:13.3f}")


if __name__ == "__main__":
    main()
