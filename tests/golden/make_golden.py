#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json from the CPU oracle (oracle/) on seeded synthetic graphs.

The reference itself cannot be run here (Boost is absent), so these vectors pin THIS repo's oracle --
they guard the oracle and the HIP path against drift and let the GPU tests check committed data.
Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from fora_amd import synth  # noqa: E402

SEED = 0x464F5241


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case(name, dangling, sources, opt):
    n, m, seed = synth.PRESETS[name]
    src, dst = synth.rmat_graph(n, m, seed, dangling)
    g = O.Graph.from_edges(n, m, src, dst)
    rmax, omega = O.fora_setting(g.n, g.m, 0.5, opt=opt)
    total, off, cnt = O.index_sizes(g, rmax, omega, opt=opt)
    rw, _, _ = O.build_index(g, SEED, rmax, omega, opt=opt)
    out = {"graph": name, "dangling": dangling, "opt": opt, "n": g.n, "m": g.m, "nnz": int(g.col.size),
           "csr_sha256": sha(g.row_ptr) + sha(g.col), "rmax": repr(rmax), "omega": repr(omega),
           "index_total": int(total), "index_sha256": sha(rw), "queries": []}
    for s in sources:
        s = int(s)
        fifo = O.push_fifo(g, s, rmax)
        N, counts = O.walk_counts(fifo, omega, opt=opt)
        ppr, res, st = O.twin_query(g, s, rmax, omega, opt=opt, seed=SEED)
        ppr_i, _, st_i = O.twin_query(g, s, rmax, omega, opt=opt, seed=SEED, index=(rw, off, cnt))
        ids, sc, rounds, _ = O.twin_topk_query(g, s, 16, 0.5, seed=SEED)
        out["queries"].append({
            "source": s,
            "fifo": {"rsum": repr(fifo["rsum"]), "pops": int(fifo["pops"]), "relax": int(fifo["relax"]),
                     "n_rw": int(N), "walks": int(counts.sum())},
            "twin": {"rsum_fix": int(st["rsum_fix"]), "levels": int(st["levels"]), "pops": int(st["pops"]),
                     "relax": int(st["relax"]), "n_walks": int(st["n_walks"]), "residue_sha256": sha(res),
                     "ppr_sha256": sha(ppr), "ppr_idx_sha256": sha(ppr_i)},
            "topk16": {"ids": ids.tolist(), "scores_hex": [float(x).hex() for x in sc], "rounds": int(rounds)} if opt else None,
        })
    return out


def main():
    rng = np.random.Generator(np.random.PCG64(99))
    cases = []
    for name, dangling, opt in (("tiny", "none", False), ("tiny", "rmat", True), ("small", "none", True)):
        n = synth.PRESETS[name][0]
        cases.append(case(name, dangling, rng.integers(0, n, size=3), opt))
    json.dump({"seed": SEED, "epsilon": 0.5, "alpha": 0.2, "cases": cases},
              open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "oracle_vectors.json"))


if __name__ == "__main__":
    main()
