#!/usr/bin/env python3
"""One-off scale check of the multi-pass bucketed push: n = 10 M, m = 100 M (1221 bins -> 2 passes per level).
Push of two sources compared bit for bit with the CPU twin; indexed query mass check.  Not part of the test suite
(about two minutes of graph generation)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import fora_amd  # noqa: E402
import oracle_lib as O  # noqa: E402
from fora_amd import synth  # noqa: E402

n, m = 10_000_000, 100_000_000
t0 = time.time()
src, dst = synth.rmat_graph(n, m, 20260110, "none")
row_ptr, col = synth.csr_from_edges(n, src, dst)
print(f"graph {n} nodes {m} edges generated in {time.time() - t0:.0f} s", flush=True)
g = O.Graph(n, m, row_ptr, col)
e = fora_amd.Engine(0)
e.set_graph(n, m, row_ptr, col)
e.set_params(epsilon=0.5, seed=7)
rmax, omega = e.get_params()
srcs = synth.query_set(n, 16, 5)
t0 = time.time()
rsv, res, st = e.push(srcs[:2])
print(f"GPU push of 2 sources: {time.time() - t0:.2f} s (first call allocates), batch={e.get_batch()}", flush=True)
for i in range(2):
    t0 = time.time()
    t = O.twin_push(g, int(srcs[i]), rmax)
    ok = bool((res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all())
    print(f"source {srcs[i]}: bit-exact vs twin = {ok}; levels {st[i]['levels']}/{t['levels']} pops {st[i]['pops']}/{t['pops']} "
          f"relax {st[i]['relax']}/{t['relax']}  (twin {time.time() - t0:.1f} s)", flush=True)
    assert ok
e.build_index()
t0 = time.time()
_, st = e.query(srcs, with_idx=True, want_ppr=False)
dt = time.time() - t0
assert all(s["ppr_sum_fix"] == 1 << 62 for s in st)
print(f"16 indexed queries: {dt:.2f} s, mass conserved exactly, idx hit = {sum(s['n_idx_hit'] for s in st) == sum(s['n_walks'] for s in st)}")
