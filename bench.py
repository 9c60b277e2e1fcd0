#!/usr/bin/env python3
"""bench.py -- SSPPR queries/sec of the HIP FORA path (BASELINE.json metric).

One "step" = one pass of the hot path (forward push + random-walk refinement) over a
batch of --queries synthetic source queries on this rank's GPU.  Default workload is
BASELINE.json configs[1]: webstanford-sized graph, eps=0.5, query_size=1000, online
walks (no index), 1 x MI355X.  N>1: one process per GPU (torch.distributed.run), the
global query list is sharded i mod N, no data-path collective (weak scaling: every
rank runs --queries queries per step).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--graph", default="webstanford", help="synth preset (webstanford|livejournal|small|tiny)")
    ap.add_argument("--dangling", default="none", choices=["none", "rmat"])
    ap.add_argument("--queries", type=int, default=1000, help="query_size per rank per step")
    ap.add_argument("--epsilon", type=float, default=0.5)
    ap.add_argument("--with-idx", action="store_true")
    ap.add_argument("--opt", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="queries in flight per launch (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--balanced", action="store_true", help="--balanced of the reference CLI (query.h:848-884), MI355X cost model")
    ap.add_argument("--balanced-start", type=float, default=1.0, help="first rmax of --balanced as a multiple of rmax (reference: 8)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="threads of the all-cores CPU leg (-1: every host core, 0: skip it)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra --balanced timing")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the L-inf check against GPU power iteration")
    ap.add_argument("--topk", type=int, default=0, help="k > 0: time `topk --opt` (config 5 style) instead of `query`")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="per-kernel FETCH_SIZE/WRITE_SIZE summary from separate rocprofv3 --pmc passes (tools/pmc_summary.py)")
    return ap.parse_args()


def cpu_baseline(g, sources, rmax, omega, args, index):
    """The oracle (FIFO push + walks in the reference's arithmetic, oracle/fora_oracle.c)
    timed on one host core over a bounded sample of the same query list."""
    import oracle_lib as O
    t0 = time.perf_counter()
    done, pops, relax, walks = 0, 0, 0, 0
    for s in sources:
        _, st = O.query(g, int(s), rmax, omega, opt=args.opt, seed=0x464F5241, index=index)
        done += 1
        pops += st["pops"]; relax += st["relax"]; walks += st["n_walks"]
        if time.perf_counter() - t0 > args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    # a few more sources push-only, to steady the algorithmic P/E estimate
    t1 = time.perf_counter()
    ppops, prelax, pn = pops, relax, done
    for s in sources[done:done + 96]:
        p = O.push_fifo(g, int(s), rmax)
        ppops += p["pops"]; prelax += p["relax"]; pn += 1
        if time.perf_counter() - t1 > 0.5 * args.cpu_seconds:
            break
    return {
        "value": done / dt, "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": f"first {done} of the {len(sources)} bench sources, oracle FIFO push + "
                  f"{'indexed' if index is not None else 'online Philox'} walks, 1 thread, {dt:.1f} s",
        "walks_per_query": walks / max(1, done),
    }, ppops / max(1, pn), prelax / max(1, pn)


def cpu_all_cores(g, sources, rmax, omega, args, index, threads):
    """the same oracle on T host threads, sources sharded, private state per call
    (ctypes releases the GIL around the C call)."""
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    budget = args.cpu_seconds

    def work(tid):
        done = 0
        for s in sources[tid::threads]:
            O.query(g, int(s), rmax, omega, opt=args.opt, seed=0x464F5241, index=index)
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        return done
    with ThreadPoolExecutor(threads) as ex:
        done = sum(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": f"{done} of the bench sources over {threads} threads (sources tid mod T), {dt:.1f} s"}


def accuracy(eng, sources, n, args, np):
    """BASELINE metric, second half: L-inf PPR error.  Exact vector = fwd_power_iteration (query.h:1192-1224,
    100 iterations) run on the GPU (fora_hip_power_iteration_batch, bit-checked against the twin in tests)."""
    ns = min(len(sources), 8 if n <= 10_000_000 else 2)
    est, _ = eng.query(sources[:ns], with_idx=args.with_idx)
    exact, _, _, _ = eng.power_iteration(sources[:ns], max_iter=100)
    err = np.abs(est - exact)
    big = exact >= 1.0 / n
    rel = float((err[big] / exact[big]).max()) if big.any() else 0.0
    return {"sources": int(ns), "linf_abs_err": float(err.max()), "max_rel_err_where_pi_ge_1_over_n": rel,
            "guarantee": f"rel err <= eps = {args.epsilon} for pi >= 1/n (algo.h:455-463)", "holds": bool(rel <= args.epsilon),
            "exact": "GPU power iteration, 100 iterations (query.h:1192-1224)"}


def main():
    args = parse()
    import torch  # first: the process must use ONE HIP runtime (torch's), the library binds to it
    import torch.distributed as dist
    import numpy as np
    import fora_amd
    from fora_amd import synth
    from fora_amd.dist import env_world, shard_sources, max_over_ranks, sum_over_ranks

    rank, local_rank, world = env_world()
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    torch.cuda.set_device(local_rank)
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
    dev = torch.device("cuda", local_rank)

    n, m, row_ptr, col = synth.preset(args.graph, args.dangling)
    eng = fora_amd.Engine(local_rank)
    arch, cus, hbm = eng.device_info()
    eng.set_graph(n, m, row_ptr, col)
    eng.set_params(alpha=0.2, epsilon=args.epsilon, opt=args.opt or bool(args.topk), seed=0x464F5241)
    rmax, omega = eng.get_params()
    if args.batch:
        eng.set_batch(args.batch)
    if args.balanced:
        eng.set_balanced(True, start_scale=args.balanced_start)
    t_idx = 0.0
    if args.with_idx:
        t0 = time.perf_counter()
        eng.build_index()
        t_idx = time.perf_counter() - t0

    # global query list, sharded i mod world; every rank gets --queries sources per step
    all_sources = synth.query_set(n, args.queries * world, 20261001)
    mine = shard_sources(all_sources, rank, world)

    topk_out = {}

    def step():
        if args.topk:
            ids, sc, rounds = eng.topk(mine, args.topk, epsilon=args.epsilon, with_idx=args.with_idx)
            topk_out["ids"], topk_out["sc"], topk_out["rounds"] = ids, sc, rounds
            return None
        _, st = eng.query(mine, with_idx=args.with_idx, want_ppr=False)
        return st

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.reset_timing()
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world if use_dist else 1, dev)
    tm = eng.timing()

    if args.topk:
        # top-k: gather the per-rank lists (RCCL all-gather when distributed), check they are sorted
        from fora_amd.dist import gather_topk
        g_ids, g_sc = gather_topk(topk_out["ids"], topk_out["sc"], len(mine) * world, rank, world if use_dist else 1,
                                  dev if use_dist else None)
        assert (np.diff(g_sc, axis=1) <= 0).all()
        if rank == 0:
            dtq = dt
            print(json.dumps({
                "metric": "SSPPR top-k queries/sec at eps=0.5", "value": len(mine) * world * args.steps / dtq,
                "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dtq / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u64 fixed-point 2^-62 (f64 at the boundary)", "data": "synthetic",
                "config": {"workload": f"{args.graph}-sized R-MAT topk k={args.topk} --opt"
                                       f"{' --with_idx' if args.with_idx else ''} query_size={args.queries}/GPU on {world}x MI355X",
                           "k": args.topk, "avg_rounds": float(np.mean(topk_out["rounds"])), "batch": eng.get_batch()},
                "phases": {k: v for k, v in tm.items() if k.endswith("_ms")}}))
        eng.close()
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    # sanity inside the bench: every query conserved mass exactly, none was skipped
    assert len(last) == len(mine)
    assert all(s["ppr_sum_fix"] == 1 << 62 for s in last), "mass not conserved"
    nd = sum(1 for s in last if not s["dangling_source"])
    tot = sum_over_ranks([len(mine), nd, tm["walks"], tm["walk_steps"], tm["relax"], tm["pops"]],
                         world if use_dist else 1, dev)

    if rank == 0:
        qps = tot[0] * args.steps / dt
        out = {
            "metric": "SSPPR queries/sec at eps=0.5", "value": qps, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 fixed-point 2^-62 (f64 at the boundary)", "data": "synthetic",
            "config": {
                "workload": f"{args.graph}-sized R-MAT (n={n}, m={m}, dangling={args.dangling}) eps={args.epsilon} "
                            f"query_size={args.queries}/GPU, fora push + "
                            f"{'indexed' if args.with_idx else 'online Philox'} walks"
                            f"{' --opt' if args.opt else ''}{' --balanced' if args.balanced else ''} on {world}x MI355X",
                "graph": args.graph, "n": n, "m": m, "epsilon": args.epsilon, "query_size_per_gpu": args.queries,
                "with_idx": bool(args.with_idx), "opt": bool(args.opt), "balanced": bool(args.balanced), "batch": eng.get_batch(),
                "sharding": f"sources i mod {world}", "non_dangling_sources": int(tot[1]),
                "rmax": rmax, "omega": omega, "device": arch, "cus": cus,
            },
        }
        q_timed = len(mine) * args.steps  # queries this rank ran in the timed region
        cpu, p_fifo, e_fifo = (None, None, None)
        if not args.no_cpu:
            import oracle_lib as O
            g = O.Graph(n, m, row_ptr, col)
            if world == 1:  # the CPU baseline is timed at N = 1 only
                index = None
                if args.with_idx:
                    rw, off, cnt = eng.get_index()
                    index = (rw, off, cnt)
                cpu, p_fifo, e_fifo = cpu_baseline(g, mine, rmax, omega, args, index)
                out["cpu_baseline"] = cpu
                threads = (os.cpu_count() or 1) if args.cpu_threads < 0 else args.cpu_threads
                if threads > 1:
                    out["cpu_baseline_all_cores"] = cpu_all_cores(g, mine, rmax, omega, args, index, threads)
            else:  # N > 1: only the algorithmic pop / relaxation counts of the FIFO oracle (push only, a few seconds)
                t1 = time.perf_counter()
                pp = pr = pn = 0
                for s in mine[:64]:
                    ps = O.push_fifo(g, int(s), rmax)
                    pp += ps["pops"]; pr += ps["relax"]; pn += 1
                    if time.perf_counter() - t1 > 5.0:
                        break
                p_fifo, e_fifo = pp / max(1, pn), pr / max(1, pn)
        if not args.no_accuracy and not args.opt:
            out["accuracy"] = accuracy(eng, mine, n, args, np)
        if world == 1 and not args.balanced and not args.no_variants:
            # the reference's other way to run the same query path (README.md:135): --balanced; not the headline value
            eng.set_balanced(True, start_scale=args.balanced_start)
            eng.query(mine, with_idx=args.with_idx, want_ppr=False)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(args.steps):
                _, stb = eng.query(mine, with_idx=args.with_idx, want_ppr=False)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - tb
            eng.set_balanced(False)
            assert all(s["ppr_sum_fix"] == 1 << 62 for s in stb)
            out["variants"] = {"balanced": {
                "value": len(mine) * args.steps / dtb, "unit": "queries/s",
                "mean_rmax_ratio": float(np.mean([s["rmax_used"] / rmax for s in stb if not s["dangling_source"]])),
                "walks_per_query": float(np.mean([s["n_walks"] for s in stb])),
                "start_scale": args.balanced_start,
                "note": "--balanced (query.h:848-884) with the MI355X cost model of fora_hip_set_balanced; same guarantee"}}
        # roofline of the push kernels: ALGORITHMIC bytes = 52 B per pop + 24 B per edge relaxation of the
        # sequential FIFO oracle; relaxations the level-synchronous schedule adds on top are
        # not credited.  Duration: HIP events around every launch.
        if tm["push_expand_launches"]:
            e_unit = e_fifo if e_fifo is not None else tm["relax"] / max(1, q_timed)
            p_unit = p_fifo if p_fifo is not None else tm["pops"] / max(1, q_timed)
            launches = tm["push_expand_launches"]
            bucketed = tm["push_accum_launches"] > 0
            fused = bucketed and tm["push_pop_launches"] == 0
            # bucketed push: one level is the kernel PAIR k_pushq_popbin + k_accum<false> (same launch count);
            # the pop is fused into the first, so the pair carries the whole push: 52 B per pop + 24 B per edge
            # relaxation of the sequential FIFO oracle, credited once against the sum of both kernels' durations
            alg_bytes = (24.0 * e_unit + (52.0 * p_unit if fused else 0.0)) * q_timed
            step_ms = tm["push_expand_ms"] + tm["push_accum_ms"]
            avg_ms = step_ms / launches
            achieved = (alg_bytes / launches) / (avg_ms * 1e-3) / 1e9
            traffic = None
            traffic_note = None
            if os.path.exists(args.traffic):
                # HBM-side bytes per launch from rocprofv3 PMC passes of this same workload (FETCH_SIZE and
                # WRITE_SIZE in separate runs, KiB -> bytes; MI355X guide: FETCH_SIZE may under-report wide
                # coalesced reads by up to 2x on gfx950 -- these kernels read 4-12 B per lane, reported raw)
                pmc = json.load(open(args.traffic))
                prefixes = (["fora::k_pushq_popbin", "fora::k_accum<false>", "fora::k_push_tail"] if fused else
                            ["fora::k_pushq_bin", "fora::k_accum<false>"] if bucketed else ["fora::k_push_expand"])
                keys = [k for k in pmc if any(k.startswith(p) for p in prefixes)]
                lead = [k for k in pmc if k.startswith(prefixes[0])]
                if lead and all("FETCH_SIZE_bytes_total" in pmc[k] for k in keys):
                    # all push kernels of the profiled batch, per level launch (the unit `achieved` uses): the bin
                    # kernel's launches plus the one k_push_tail that finishes the small levels
                    n_launch = sum(pmc[k].get("FETCH_SIZE_dispatches", 0) for k in keys
                                   if k.startswith(prefixes[0]) or k.startswith("fora::k_push_tail"))
                    traffic = sum(pmc[k].get("FETCH_SIZE_bytes_total", 0) + pmc[k].get("WRITE_SIZE_bytes_total", 0)
                                  for k in keys) / max(1, n_launch)
                    traffic_note = pmc.get("_note")
            by_kernel = {("k_pushq_popbin" if fused else "k_pushq_bin" if bucketed else "k_push_expand"): tm["push_expand_ms"] / launches}
            if bucketed:
                by_kernel["k_accum<false>"] = tm["push_accum_ms"] / max(1, tm["push_accum_launches"])
            if tm["push_pop_launches"]:
                by_kernel["k_pushq_pop" if bucketed else "k_push_pop"] = tm["push_pop_ms"] / tm["push_pop_launches"]
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                "kernel": ("fora::k_pushq_popbin + fora::k_accum<false> (one level of the push; k_push_tail finishes the small levels)" if fused else
                           "fora::k_pushq_bin + fora::k_accum<false> (expand step of one level)" if bucketed
                           else "fora::k_push_expand"),
                "launches": int(launches), "avg_launch_ms": avg_ms, "avg_ms_by_kernel": by_kernel,
                "algorithmic_bytes_per_launch": alg_bytes / launches,
                "algorithmic_bytes": "52 B per pop + 24 B per edge relaxation of the sequential FIFO oracle" if fused
                                     else "24 B per edge relaxation of the sequential FIFO oracle",
                "fifo_relaxations_per_query": e_unit, "fifo_pops_per_query": p_unit,
                "gpu_relaxations_per_query": tm["relax"] / max(1, q_timed),
                "push_total": {  # all push kernels against 52*P + 24*E
                    "achieved": (52.0 * p_unit + 24.0 * e_unit) * q_timed
                                / ((step_ms + tm["push_pop_ms"]) * 1e-3) / 1e9,
                    "ms": step_ms + tm["push_pop_ms"]},
            }
        walk_bytes = (tm["walk_steps"] * 20.0 + tm["walks"] * 16.0) if not args.with_idx else tm["walks"] * 20.0
        out["phases"] = {
            "push_pop_ms": tm["push_pop_ms"], "push_expand_ms": tm["push_expand_ms"], "push_accum_ms": tm["push_accum_ms"],
            "walk_alloc_ms": tm["walk_alloc_ms"], "walk_ms": tm["walk_ms"], "walk_accum_ms": tm["walk_accum_ms"], "other_ms": tm["other_ms"],
            "batch_ms": tm["batch_ms"], "levels_launched": tm["levels"],
            "walks": tm["walks"], "walk_steps": tm["walk_steps"],
            "walks_per_s": tm["walks"] / max(1e-9, tm["walk_ms"] * 1e-3),
            "walk_algorithmic_GBps": walk_bytes / max(1e-9, tm["walk_ms"] * 1e-3) / 1e9,
            "index_build_s": t_idx,
        }
        print(json.dumps(out))
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
