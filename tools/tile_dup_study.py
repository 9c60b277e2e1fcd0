#!/usr/bin/env python3
"""Experiment tooling (VERDICT r05 #1b, "count duplicate targets per tile first"): how many of the messages of a wide-layout bin tile
name a (slot, target) pair that the same tile names again?  A numpy emulation of the level-synchronous push (f64: counting only)
on a preset graph; per level the frontier in node order is cut into tiles of `tile` entries exactly as k_pushq_bin does (granules of 64
entries from tile/64 equal segments of the list), the targets of a tile's rows are counted with and without the hub nodes that the
kernels sum in LDS anyway.  usage: python tools/tile_dup_study.py <graph> [sources] [tile] [hubs]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fora_amd import synth  # noqa: E402

graph = sys.argv[1] if len(sys.argv) > 1 else "medium"
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 512
hubs = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
n, m, rp, col = synth.preset(graph)
deg = np.diff(rp).astype(np.int64)
indeg = np.bincount(col, minlength=n)
is_hub = np.zeros(n, dtype=bool)
if hubs:
    is_hub[np.argsort(-indeg, kind="stable")[:hubs]] = True
alpha, eps = 0.2, 0.5
delta, pfail = 1.0 / n, 1.0 / n
rmax = eps * np.sqrt(delta / 3 / m / np.log(2 / pfail))
thr = rmax * np.maximum(deg, 0)
srcs = [int(s) for s in synth.query_set(n, 64, 20261001) if deg[s] > 0][:nsrc]
tot_msgs = tot_dup = tot_msgs_nh = tot_dup_nh = 0
GR = 64
for s in srcs:
    res = np.zeros(n)
    res[s] = 1.0
    front = np.array([s], dtype=np.int64)
    level = 0
    while front.size and level < 200:
        r = res[front].copy()
        res[front] = 0
        live = deg[front] > 0
        inc = np.where(live, (1 - alpha) * r / np.maximum(deg[front], 1), 0.0)
        dang = ((1 - alpha) * r[~live]).sum()
        # tiles as in k_pushq_bin (tile_pos): ntiles tiles, NT / GR granules each, granule g of tile t = entries [g * seg + t * GR, + GR)
        cnt = front.size
        ntiles = (cnt + tile - 1) // tile
        seg = ntiles * GR
        pos = np.arange(ntiles * tile, dtype=np.int64)
        t_of, own = pos // tile, pos % tile
        fpos = (own // GR) * seg + t_of * GR + own % GR
        ok = fpos < cnt
        fp, tid = fpos[ok], t_of[ok]
        rows = front[fp]
        lens = deg[rows]
        tt = np.repeat(tid, lens)
        starts = np.repeat(rp[rows], lens)
        offs = np.arange(lens.sum(), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        tgt = col[starts + offs].astype(np.int64)
        key = tt * n + tgt
        uniq = np.unique(key).size
        tot_msgs += key.size
        tot_dup += key.size - uniq
        nh = ~is_hub[tgt]
        tot_msgs_nh += int(nh.sum())
        tot_dup_nh += int(nh.sum()) - np.unique(key[nh]).size
        # the level's adds
        np.add.at(res, tgt, np.repeat(inc[fp], lens))
        if dang:
            res[s] += dang
        front = np.flatnonzero((res >= thr) & (res > 0))
        level += 1
print(json.dumps({"graph": graph, "n": n, "m": m, "sources": srcs, "tile_entries": tile, "hubs_excluded": hubs,
                  "messages": int(tot_msgs), "duplicate_share": tot_dup / max(1, tot_msgs),
                  "messages_without_hubs": int(tot_msgs_nh), "duplicate_share_without_hubs": tot_dup_nh / max(1, tot_msgs_nh)}))
