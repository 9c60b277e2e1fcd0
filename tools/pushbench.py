#!/usr/bin/env python3
"""Times the push kernels of one or more builds of libfora_hip.so on the headline workload (ws-sized graph, 1000 sources).
Experiment tooling: `python tools/pushbench.py [--graph G] [--queries Q] [--reps R] [--mode push|query|idx] lib1.so lib2.so ...`
Each library runs in its own process (FORA_HIP_LIB selects it); prints one line per library."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import fora_amd
    from fora_amd import synth
    n, m, row_ptr, col = synth.preset(args.graph, args.dangling)
    eng = fora_amd.Engine(0)
    eng.set_graph(n, m, row_ptr, col)
    eng.set_params(alpha=0.2, epsilon=0.5, seed=0x464F5241)
    if args.batch:
        eng.set_batch(args.batch)
    for kv in args.option:  # engine knobs the environment does not reach (schedule knobs: rounds, round_div, defer, defer_min)
        name, value = kv.split("=")
        eng.set_option(name, int(value))
    srcs = synth.query_set(n, args.queries, 20261001)
    if args.mode == "idx":
        eng.build_index()

    def run():
        if args.mode == "push":
            return eng.push(srcs, want=False)
        return eng.query(srcs, with_idx=args.mode == "idx", want_ppr=False)[1]
    st = run()
    eng.reset_timing()
    for _ in range(args.reps):
        st = run()
    tm = eng.timing()
    R = args.reps
    lv = max(1, tm["push_expand_launches"])
    out = {"lib": os.path.basename(os.environ.get("FORA_HIP_LIB", "libfora_hip.so")),
           "bin_ms": tm["push_expand_ms"] / R, "accum_ms": tm["push_accum_ms"] / R,
           "tail_ms": tm["push_tail_ms"] / R, "team_ms": tm["push_team_ms"] / R,
           "push_ms": (tm["push_expand_ms"] + tm["push_accum_ms"] + tm["push_pop_ms"] + tm["push_tail_ms"] + tm["push_team_ms"]) / R,
           "bin_avg": tm["push_expand_ms"] / lv, "accum_avg": tm["push_accum_ms"] / max(1, tm["push_accum_launches"]),
           "launches": lv / R, "walk_alloc_ms": tm["walk_alloc_ms"] / R, "walk_ms": tm["walk_ms"] / R,
           "walk_accum_ms": tm["walk_accum_ms"] / R, "other_ms": tm["other_ms"] / R, "batch_ms": tm["batch_ms"] / R,
           "relax_per_q": tm["relax"] / R / len(srcs), "pops_per_q": tm["pops"] / R / len(srcs),
           "walks_per_q": tm["walks"] / R / len(srcs), "batch": eng.get_batch(),
           "rsum_fix_xor": int(np.bitwise_xor.reduce(np.asarray([int(s["rsum_fix"]) for s in st], dtype=np.uint64)))}
    stp = eng.stamps()
    if stp.any():
        out["stamps_bin_Mcyc"] = [round(int(x) / 1e6 / R, 1) for x in stp[:10]]
        out["stamps_acc_Mcyc"] = [round(int(x) / 1e6 / R, 1) for x in stp[16:22]]
        out["stamps_raw4"] = int(stp[4]) // R
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--graph", default="webstanford")
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--mode", default="push", choices=["push", "query", "idx"])
    ap.add_argument("--dangling", default="none", choices=["none", "rmat"])
    ap.add_argument("--option", action="append", default=[], help="name=value for fora_hip_set_option (repeatable)")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    libs = args.libs or [os.path.join(ROOT, "fora_amd", "libfora_hip.so")]
    for lib in libs:
        env = dict(os.environ, FORA_HIP_LIB=os.path.abspath(lib))
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--graph", args.graph, "--queries", str(args.queries),
               "--reps", str(args.reps), "--batch", str(args.batch), "--mode", args.mode, "--dangling", args.dangling] + [x for kv in args.option for x in ("--option", kv)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode:
            sys.stdout.write(f"{lib}: rc={r.returncode} {r.stderr[-800:]}\n")


if __name__ == "__main__":
    main()
