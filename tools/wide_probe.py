#!/usr/bin/env python3
"""Experiment tooling (GPU box): the push of a wide-layout graph through one build of libfora_hip.so (FORA_HIP_LIB),
push only, one warm-up + `reps` timed calls.  Prints one JSON line: kernel milliseconds, relaxations, CU cycles per relaxation.
usage: FORA_HIP_LIB=variants/lib_x.so python tools/wide_probe.py <graph> <queries> [reps] [name=value ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fora_amd  # noqa: E402
from fora_amd import synth  # noqa: E402

graph, nq = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 1
opts = [a for a in sys.argv[3:] if "=" in a]
n, m, rp, col = synth.preset(graph)
eng = fora_amd.Engine(0)
for kv in opts:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
eng.set_graph(n, m, rp, col)
eng.set_params(alpha=0.2, epsilon=0.5, seed=0x464F5241)
srcs = synth.query_set(n, nq, 20261001)
st = eng.push(srcs, want=False)
eng.reset_timing()
for _ in range(reps):
    st = eng.push(srcs, want=False)
tm = eng.timing()
x = 0
relax = 0
for s in st:
    x ^= int(s["rsum_fix"]) ^ int(s["relax"])
    relax += int(s["relax"])
CUS, GHZ = 256, 2.4
out = {"graph": graph, "lib": os.path.basename(os.environ.get("FORA_HIP_LIB", "libfora_hip.so")), "opts": opts, "queries": nq, "xor": x,
       "bin_ms": round(tm["push_expand_ms"] / reps, 2), "accum_ms": round(tm["push_accum_ms"] / reps, 2), "tail_ms": round(tm["push_tail_ms"] / reps, 2),
       "launches": tm["push_expand_launches"] / reps, "relax": relax, "batch": eng.get_batch(),
       "bin_cyc_per_relax": round(tm["push_expand_ms"] / reps * 1e-3 * CUS * GHZ * 1e9 / relax, 3),
       "acc_cyc_per_relax": round(tm["push_accum_ms"] / reps * 1e-3 * CUS * GHZ * 1e9 / relax, 3)}
stp = eng.stamps()
if stp.any():
    out["stamps_bin_Mcyc"] = [round(int(v) / 1e6 / (reps + 1), 1) for v in stp[:8]]
    out["stamps_acc_Mcyc"] = [round(int(v) / 1e6 / (reps + 1), 1) for v in stp[16:22]]
print(json.dumps(out), flush=True)
