#!/usr/bin/env python3
"""Builder run (GPU box): pops / relaxations of the sequential FIFO oracle against the GPU schedule's on a few sources of
a large preset, for graphs where one FIFO push is too slow for the default bench run (Twitter-2010-sized: about a minute
of one core).  bench.py credits such graphs with GPU counts x these ratios and says so in `algorithmic_counts`.
usage: python3 tools/fifo_counts.py twitter2010 [nsources=2]  ->  gpurun_out/fifo_counts_<graph>.json (copy to profiles/)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import fora_amd
    import oracle_lib as O
    from fora_amd import synth
    graph = sys.argv[1] if len(sys.argv) > 1 else "twitter2010"
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n, m, row_ptr, col = synth.preset(graph)
    eng = fora_amd.Engine(0)
    eng.set_graph(n, m, row_ptr, col)
    eng.set_params(alpha=0.2, epsilon=0.5, seed=0x464F5241)
    rmax, omega = eng.get_params()
    srcs = synth.query_set(n, 1000, 20261001)
    srcs = np.array([s for s in srcs if row_ptr[s + 1] > row_ptr[s]][:ns], dtype=np.int32)
    st = eng.push(srcs, want=False)
    gp = sum(int(x["pops"]) for x in st)
    gr = sum(int(x["relax"]) for x in st)
    eng.close()
    g = O.Graph(n, m, row_ptr, col)
    fp = fr = 0
    t0 = time.perf_counter()
    for s in srcs:
        p = O.push_fifo(g, int(s), rmax)
        fp += p["pops"]; fr += p["relax"]
    dt = time.perf_counter() - t0
    out = {"graph": graph, "n": n, "m": m, "sources": [int(s) for s in srcs], "fifo_pops": fp, "fifo_relax": fr, "gpu_pops": gp,
           "gpu_relax": gr, "fifo_seconds": dt,
           "note": f"profiles/fifo_counts_{graph}.json: FIFO oracle vs GPU schedule on the first {len(srcs)} non-dangling bench "
                   f"sources (tools/fifo_counts.py, {dt:.0f} s of one core)"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"fifo_counts_{graph}.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
