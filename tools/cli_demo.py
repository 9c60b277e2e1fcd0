#!/usr/bin/env python3
"""Writes a webstanford-sized dataset folder (attribute.txt / graph.txt / ssquery.txt) and runs the
`fora` command line on it the way the reference is run (README of wangsibovictor/fora):
  fora build, fora query, fora query --with_idx, fora topk --opt --with_idx --k 500
usage: cli_demo.py <workdir> [query_size]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from fora_amd import synth  # noqa: E402

work = sys.argv[1]
qs = sys.argv[2] if len(sys.argv) > 2 else "1000"
folder = os.path.join(work, "data", "webstanford")
os.makedirs(folder, exist_ok=True)
n, m, seed = synth.PRESETS["webstanford"]
t0 = time.time()
src, dst = synth.rmat_graph(n, m, seed, "none")
open(os.path.join(folder, "attribute.txt"), "w").write(f"n={n}\nm={m}\n")
np.savetxt(os.path.join(folder, "graph.txt"), np.stack([src, dst], 1), fmt="%d")
np.savetxt(os.path.join(folder, "ssquery.txt"), synth.query_set(n, 1000, 20261001), fmt="%d")
print(f"# dataset written in {time.time() - t0:.1f} s", flush=True)
fora = os.path.join(ROOT, "fora_amd", "bin", "fora")
common = ["--prefix", os.path.join(work, "data") + "/", "--dataset", "webstanford", "--epsilon", "0.5",
          "--result_dir", os.path.join(work, "res")]
for args in (["build"], ["build", "--opt"],
             ["query", "--algo", "fora", "--query_size", qs],
             ["query", "--algo", "fora", "--query_size", qs, "--with_idx"],
             ["topk", "--algo", "fora", "--opt", "--with_idx", "--k", "500", "--query_size", qs]):
    t0 = time.time()
    r = subprocess.run([fora, *args, *common], capture_output=True, text=True)
    keep = [l for l in r.stdout.splitlines() if "source node" not in l and not l.startswith("---")]
    print(f"$ fora {' '.join(args)} ...   (rc={r.returncode}, {time.time() - t0:.2f} s wall incl. text parse)")
    print("\n".join("    " + l for l in keep[-14:]))
    if r.returncode:
        print(r.stderr[-500:])
for f in sorted(os.listdir(os.path.join(work, "res", "execution"))):
    print("# result file:", f)
