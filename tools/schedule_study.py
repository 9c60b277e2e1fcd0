#!/usr/bin/env python3
"""Experiment tooling (CPU): what a push schedule costs on the ws-sized bench graph, counted with the twin -- pops, edge
relaxations, levels by size, rsum (= walks) -- for plain levels and for bounded deferral (twin_set_defer / _defer_min),
next to the sequential FIFO of algo.h:980-1017.  `python tools/schedule_study.py [--sources 6]`"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="webstanford")
    ap.add_argument("--sources", type=int, default=6)
    args = ap.parse_args()
    import oracle_lib as orc
    from fora_amd import synth
    orc.build()
    n, m, seed = synth.PRESETS[args.graph]
    t0 = time.time()
    src, dst = synth.rmat_graph(n, m, seed, "none")
    g = orc.Graph.from_edges(n, m, src, dst)
    rmax, omega = orc.fora_setting(n, m, 0.5)
    srcs = synth.query_set(n, 1000, 20261001)[:args.sources]
    print(f"graph {args.graph} n={n} m={m} built in {time.time() - t0:.1f}s rmax={rmax:.4g}", flush=True)
    fifo = [orc.push_fifo(g, int(s), rmax) for s in srcs]
    fp = np.mean([f[2]["pops"] if isinstance(f, tuple) else f["pops"] for f in fifo])
    fr = np.mean([f[2]["relax"] if isinstance(f, tuple) else f["relax"] for f in fifo])
    frs = np.mean([f[2]["rsum"] if isinstance(f, tuple) else f["rsum"] for f in fifo])
    print(f"FIFO: pops {fp:.0f} relax {fr:.0f} rsum {frs:.4f}")
    for k, dmin in [(0, 0), (1, 0), (1, 4097), (1, 16384), (1, 32768), (2, 4097), (2, 16384), (3, 4097)]:
        orc.twin_set_defer(k)
        orc.twin_set_defer_min(dmin)
        rows = [orc.twin_push(g, int(s), rmax) for s in srcs]
        pops = np.mean([r["pops"] for r in rows]); relax = np.mean([r["relax"] for r in rows])
        rsum = np.mean([r["rsum_fix"] / 2.0**62 for r in rows])
        lv = [r["level_sizes"] for r in rows]
        big = np.mean([(x > 4096).sum() for x in lv]); mid = np.mean([((x > 512) & (x <= 4096)).sum() for x in lv]); alll = np.mean([x.size for x in lv])
        print(f"defer {k} min {dmin:6d}: pops {pops:9.0f} ({pops / fp:.3f}x) relax {relax:9.0f} ({relax / fr:.3f}x) rsum {rsum:.4f} ({rsum / frs:.3f}x) "
              f"levels >4096: {big:.1f}  512..4096: {mid:.1f}  all: {alll:.1f}", flush=True)
    orc.twin_set_defer(0); orc.twin_set_defer_min(0)


if __name__ == "__main__":
    main()
