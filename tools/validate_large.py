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

# usage: validate_large.py [n m [dangling [nq]]]   (default 10 M / 100 M; Twitter-2010 size: 41652230 1468365182 rmat 16)
n = int(sys.argv[1]) if len(sys.argv) > 2 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
dangling = sys.argv[3] if len(sys.argv) > 3 else "none"
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 16
t0 = time.time()
src, dst = synth.rmat_graph(n, m, 20260110, dangling, fold=True)
row_ptr, col = synth.csr_from_edges(n, src, dst)
print(f"graph {n} nodes {m} edges generated in {time.time() - t0:.0f} s", flush=True)
g = O.Graph(n, m, row_ptr, col)
e = fora_amd.Engine(0)
e.set_graph(n, m, row_ptr, col)
e.set_params(epsilon=0.5, seed=7)
rmax, omega = e.get_params()
deg = np.diff(row_ptr)
cand = synth.query_set(n, 64 * nq, 5)
srcs = cand[deg[cand] > 0][:nq].copy()  # non-dangling sources
t0 = time.time()
rsv, res, st = e.push(srcs[:2])
print(f"GPU push of 2 sources: {time.time() - t0:.2f} s (first call allocates), batch={e.get_batch()}", flush=True)
for i in range(1 if n > 20_000_000 else 2):
    t0 = time.time()
    t = O.twin_push(g, int(srcs[i]), rmax)
    ok = bool((res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all())
    print(f"source {srcs[i]}: bit-exact vs twin = {ok}; levels {st[i]['levels']}/{t['levels']} pops {st[i]['pops']}/{t['pops']} "
          f"relax {st[i]['relax']}/{t['relax']}  (twin {time.time() - t0:.1f} s)", flush=True)
    assert ok
t0 = time.time()
e.build_index()
total, _, _ = e.index_sizes()
print(f"index: {total} walks built in {time.time() - t0:.2f} s", flush=True)
_, st = e.query(srcs, with_idx=True, want_ppr=False)  # warm-up: allocates the workspace for this batch size
e.reset_timing()
t0 = time.time()
_, st = e.query(srcs, with_idx=True, want_ppr=False)
dt = time.time() - t0
assert all(s["ppr_sum_fix"] == 1 << 62 for s in st)
tm = e.timing()
print(f"{len(srcs)} indexed queries: {dt:.2f} s = {len(srcs) / dt:.2f} queries/s, batch={e.get_batch()}, mass conserved exactly, "
      f"idx hit = {sum(s['n_idx_hit'] for s in st) == sum(s['n_walks'] for s in st)}")
print("per query: relax %.3g walks %.3g levels %.0f; phases ms:" % (tm["relax"] / len(srcs), tm["walks"] / len(srcs), tm["levels"]),
      {k: round(v, 1) for k, v in tm.items() if k.endswith("_ms")})

# config 5 style: top-k (k = 500, --opt, --with_idx) on the same graph
if len(sys.argv) > 5 and sys.argv[5] == "topk":
    e.clear_index()
    e.set_params(epsilon=0.5, opt=True, seed=7)
    t0 = time.time()
    e.build_index()
    total, _, _ = e.index_sizes()
    print(f"--opt index: {total} walks built in {time.time() - t0:.2f} s", flush=True)
    nk = min(len(srcs), 8)
    e.topk(srcs[:nk], 500, epsilon=0.5, with_idx=True)  # warm-up with the same batch size (no re-allocation when timed)
    e.reset_timing()
    t0 = time.time()
    ids, sc, rounds = e.topk(srcs[:nk], 500, epsilon=0.5, with_idx=True)
    dt = time.time() - t0
    tm = e.timing()
    assert (np.diff(sc, axis=1) <= 0).all() and (sc[:, 0] > 0).all()
    print(f"{nk} top-k queries (k=500 --opt --with_idx): {dt:.2f} s = {nk / dt:.2f} queries/s, rounds {rounds.tolist()}, "
          f"idx hit ratio {tm['idx_hits'] / max(1, tm['walks']):.3f}")
    print("phases ms:", {k: round(v, 1) for k, v in tm.items() if k.endswith("_ms")})
