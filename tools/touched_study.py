#!/usr/bin/env python3
"""Experiment tooling (GPU box): how much of a slot's slabs a top-k round touches on the Twitter-2010-sized graph -- nodes with
residue / reserve after a push at the thresholds of the --opt top-k rounds (algo.h:466-474), and the share of the slabs'
64-byte lines (8 nodes) and 4-KiB pages that hold at least one of them."""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import fora_amd
    from fora_amd import synth
    name = sys.argv[1] if len(sys.argv) > 1 else "twitter2010"
    k, eps = 500, 0.5
    t0 = time.time()
    n, m, row_ptr, col = synth.preset(name)
    eng = fora_amd.Engine(0)
    eng.set_graph(n, m, row_ptr, col)
    print(f"{name}: n={n} m={m} loaded in {time.time() - t0:.1f}s", flush=True)
    srcs = synth.query_set(n, 64, 20261001)
    deg = np.diff(row_ptr)
    srcs = np.array([s for s in srcs if deg[s] > 0][:3], dtype=np.int32)
    delta = 1.0 / k / 10
    pfail = 1.0 / n / n
    for rnd in range(1, 6):
        rmax = eps * math.sqrt(delta / 3 / m / math.log(2 / pfail))
        rmax *= math.sqrt(1.0 * m * rmax) * 3
        eng.set_params_raw(alpha=0.2, rmax=rmax, omega=1.0, seed=1)
        rsv, res, st = eng.push(srcs)
        for i in range(len(srcs)):
            touched = (res[i] != 0) | (rsv[i] != 0)
            nz = int(touched.sum())
            lines = touched[: n // 8 * 8].reshape(-1, 8).any(axis=1).mean()
            pages = touched[: n // 512 * 512].reshape(-1, 512).any(axis=1).mean()
            print(f"round {rnd} delta {delta:.3g} rmax {rmax:.3g} src {srcs[i]}: pops {st[i]['pops']} touched {nz} ({nz / n:.4%}) lines {lines:.3%} 4K-pages {pages:.3%}", flush=True)
        delta = max(1.0 / n, delta / 4)
    eng.close()


if __name__ == "__main__":
    main()
