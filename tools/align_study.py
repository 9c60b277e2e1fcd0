#!/usr/bin/env python3
"""Experiment tooling (GPU box): upper bound of what aligning the slots' frontier lists could give the slot-major bin kernel -- a batch whose slots all
run the SAME source has identical lists (every tile of every slot covers the same nodes) against a batch of distinct sources, in both dispatch orders.
usage: python tools/align_study.py [graph] [slots]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fora_amd  # noqa: E402
from fora_amd import synth  # noqa: E402

graph = sys.argv[1] if len(sys.argv) > 1 else "livejournal"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 143
n, m, rp, col = synth.preset(graph)
eng = fora_amd.Engine(0)
eng.set_graph(n, m, rp, col)
eng.set_params(alpha=0.2, epsilon=0.5, seed=0x464F5241)
eng.build_index()
deg = np.diff(rp)
distinct = np.array([s for s in synth.query_set(n, 4 * nq, 20261001) if deg[s] > 0][:nq], dtype=np.int32)
same = np.full(nq, distinct[1], dtype=np.int32)
for name, srcs in (("distinct", distinct), ("same", same)):
    for sm in (0, 5, 0, 5):
        eng.set_option("slot_major", sm)
        eng.query(srcs, with_idx=True, want_ppr=False)
        eng.reset_timing()
        for _ in range(2):
            eng.query(srcs, with_idx=True, want_ppr=False)
        tm = eng.timing()
        print(json.dumps({"graph": graph, "sources": name, "slot_major": sm, "slots": nq, "bin_ms": round(tm["push_expand_ms"] / 2, 1), "accum_ms": round(tm["push_accum_ms"] / 2, 1),
                          "walk_ms": round(tm["walk_ms"] / 2, 1), "batch_ms": round(tm["batch_ms"] / 2, 1), "relax": tm["relax"] // 2}), flush=True)
