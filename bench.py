#!/usr/bin/env python3
"""bench.py -- SSPPR queries/sec of the HIP FORA path (BASELINE.json metric).

One "step" = one pass of the hot path (forward push + random-walk refinement, or the top-k driver with --topk)
over a batch of synthetic source queries.  Workloads (BASELINE.json configs):

  default                                   config 2: webstanford-sized, eps=0.5, query_size=1000, online walks, 1 GPU
  --graph livejournal --with-idx            config 3: LiveJournal-sized, indexed walks in HBM
  --graph twitter2010 --with-idx --gpus 8 --scaling strong --queries 1000
                                            config 4: Twitter-2010-sized, 1000 sources sharded i mod 8
  --graph twitter2010 --topk 500 --with-idx --gpus 8 --scaling strong --queries 1000
                                            config 5: top-k (k=500, --opt), RCCL all-gather of the lists inside the timed region

--gpus N: one process per GPU.  Started by `python -m torch.distributed.run` (the driver's way) this file is a rank;
started plainly with --gpus N > 1 it launches that same command itself, before anything touches a GPU.
--scaling weak (default): every rank runs --queries sources per step; strong: --queries sources in total, source i on
rank i mod N (query.h:1471-1476 has no cross-query state).  No data-path collective for `query`; `topk` ends every
step with one all-gather of fixed-size [ceil(Q/N), k] (int32 id, f64 score) lists.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
DEFAULT_QUERIES = {"twitter2010": 125}  # one GPU's share of config 4 (1000 sources over 8 GPUs); everything else 1000
DTYPE = "u64 fixed-point 2^-62 (f64 at the boundary)"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--graph", default="webstanford", help="synth preset (webstanford|medium|livejournal|twitter2010|small|tiny)")
    ap.add_argument("--dangling", default="none", choices=["none", "rmat"])
    ap.add_argument("--queries", type=int, default=0, help="query_size (per rank with --scaling weak, in total with strong); 0: per graph")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--epsilon", type=float, default=0.5)
    ap.add_argument("--with-idx", action="store_true")
    ap.add_argument("--opt", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="queries in flight per launch (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--balanced", action="store_true", help="--balanced of the reference CLI (query.h:848-884), MI355X cost model")
    ap.add_argument("--balanced-start", type=float, default=1.0, help="first rmax of --balanced as a multiple of rmax (reference: 8)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--fifo-sample-seconds", type=float, default=45.0, help="... and keep sampling until this many seconds are spent")
    ap.add_argument("--fifo-sample", type=int, default=0,
                    help="graphs whose FIFO push is too slow for the CPU legs: run this many sources through the FIFO oracle anyway (push only) "
                         "so that the roofline fraction on in-run counts is printed beside the one on the builder's ratio file")
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="threads of the all-cores CPU leg (-1: every host core, 0: skip it)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra --balanced timing")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the L-inf check against exact PPR")
    ap.add_argument("--no-configs", action="store_true", help="default run only: skip the brief measurements of BASELINE configs 3-5")
    ap.add_argument("--topk", type=int, default=0, help="k > 0: time `topk --opt` (config 5 style) instead of `query`")
    ap.add_argument("--traffic", default="", help="per-kernel FETCH_SIZE/WRITE_SIZE summary from separate rocprofv3 --pmc passes "
                                                  "(tools/pmc_summary.py); default profiles/pmc_traffic_<graph>[_idx].json")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="test hook (tests/test_dist_cpu.py): launcher, sharding, barrier / max-over-ranks timing and the top-k "
                         "gather with gloo on CPU and a stand-in step; no GPU, no engine")
    args = ap.parse_args(argv)
    if not args.queries:
        args.queries = DEFAULT_QUERIES.get(args.graph, 1000)
    return args


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args):
    """--gpus N > 1 outside torch.distributed.run: start N ranks of this file (fresh children; this process has not
    touched a GPU and never will) and pass their exit code on."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def my_sources(np, synth, n, args, rank, world):
    """Global query list and this rank's shard.  weak: --queries per rank; strong: --queries in total.  Source i of the
    global list runs on rank i mod world either way."""
    from fora_amd.dist import shard_sources
    total = args.queries * world if args.scaling == "weak" else args.queries
    all_sources = synth.query_set(n, total, 20261001)
    return all_sources, shard_sources(all_sources, rank, world)


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
        if time.perf_counter() - t1 > 0.5 * args.cpu_seconds:
            break
        p = O.push_fifo(g, int(s), rmax)
        ppops += p["pops"]; prelax += p["relax"]; pn += 1
    return {
        "value": done / dt, "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": f"first {done} of the {len(sources)} bench sources, oracle FIFO push + "
                  f"{'indexed' if index is not None else 'online Philox'} walks, 1 thread, {dt:.1f} s",
        "walks_per_query": walks / max(1, done),
    }, ppops, prelax, pn  # FIFO pops / relaxations summed over the first pn sources


def host_cpus():
    """What this process may really use of the host: the CPUs of its affinity mask (not os.cpu_count(), which counts the
    machine's), the physical cores behind them (hardware threads that share a core counted once) and the cgroup's CPU
    quota if there is one."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = {}
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        cores.setdefault(sib, []).append(c)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # one CPU per physical core first, then the sibling threads: T threads bind to the first T entries
    order = [v[0] for v in cores.values()] + [c for v in cores.values() for c in v[1:]]
    usable = len(allowed) if quota is None else max(1, min(len(allowed), int(quota)))
    return {"hardware_threads_allowed": len(allowed), "physical_cores_allowed": len(cores), "cgroup_cpu_quota": quota,
            "threads_usable": usable, "machine_hardware_threads": os.cpu_count(), "model": model, "bind_order": order}


def cpu_all_cores(g, sources, rmax, omega, args, index, threads, host, one_thread_value=None):
    """the same oracle on T host threads, sources t, t + T, ... per thread -- pthreads inside the
    oracle (orc_query_many_pinned), every thread bound to one CPU of the process's affinity mask (physical cores first)
    and with its own buffers, allocated after binding, for the whole run.  Also a short scaling row (T = 1, 8, 64, ...)
    so that the all-core figure can be read against the single-thread one."""
    import oracle_lib as O
    order = host["bind_order"]
    rows = []
    ladder = sorted({t for t in (1, 8, 64, host["physical_cores_allowed"], threads) if 1 <= t <= threads})
    budget = max(2.0, args.cpu_seconds / 3.0)
    for t in ladder:
        done, dt, _ = O.query_many(g, sources, rmax, omega, t, budget if t != threads else args.cpu_seconds, opt=args.opt,
                                   seed=0x464F5241, index=index, cpus=order[:max(1, min(t, len(order)))])
        rows.append({"threads": t, "queries": done, "seconds": dt, "value": done / dt})
    top = rows[-1]
    # the T = 1 rung of the ladder runs for a third of the budget over a handful of sources: noise.  The speed-up is quoted
    # against the one-thread baseline of this same run (cpu_baseline: the full budget, the same source order)
    one = one_thread_value if one_thread_value else (rows[0]["value"] if rows[0]["threads"] == 1 else None)
    return {"value": top["value"], "unit": "queries/s", "cores": min(host["physical_cores_allowed"], threads), "threads": threads, "kind": "port",
            "sample": f"{top['queries']} of the bench sources over {threads} pinned pthreads (sources tid mod T, private buffers), {top['seconds']:.1f} s",
            "scaling": rows, "scaling_note": "rungs below the top one run for a third of the budget each (`queries` = sources they finished): a trend, not a measurement",
            "speedup_over_one_thread": (top["value"] / one) if one else None,
            "speedup_reference": "cpu_baseline.value of this run (1 thread, full budget)" if one_thread_value else "first rung of `scaling`",
            "host": {k: v for k, v in host.items() if k != "bind_order"},
            "note": "every thread runs whole queries (a webstanford-sized query is 6.5 M scattered 8-byte read-modify-writes and 16 M "
                    "dependent walk steps over ~11 MB of private arrays); cores = physical cores the threads run on (one pinned thread per core up to the "
                    "cgroup's CPU quota), threads = pthreads started; host.* = what the box has and what the process may use"}


def accuracy(eng, g, sources, n, m, args, np):
    """BASELINE metric, second half: L-inf PPR error against exact PPR = fwd_power_iteration (query.h:1192-1224, 100
    iterations).  The first sources are checked against the CPU restatement of that function in f64
    (oracle/fora_oracle.c orc_power_iteration -- independent of the HIP library), the rest against the same iteration
    run on the GPU (fora_hip_power_iteration_batch, bit-checked against the twin in tests)."""
    ns = min(len(sources), 8 if n <= 10_000_000 else 2)
    n_cpu = 0 if g is None else (2 if m <= 20_000_000 else 1 if m <= 200_000_000 else 0)  # 100 sweeps of m edges on one core
    n_cpu = min(n_cpu, ns)
    est, _ = eng.query(sources[:ns], with_idx=args.with_idx)
    exact, _, _, _ = eng.power_iteration(sources[:ns], max_iter=100)
    cpu_vs_gpu_exact = None
    if n_cpu:
        import oracle_lib as O
        cpu_exact = np.stack([O.power_iteration(g, int(s)) for s in sources[:n_cpu]])
        cpu_vs_gpu_exact = float(np.abs(cpu_exact - exact[:n_cpu]).max())
        exact[:n_cpu] = cpu_exact
    err = np.abs(est - exact)
    big = exact >= 1.0 / n
    rel = float((err[big] / exact[big]).max()) if big.any() else 0.0
    return {"sources": int(ns), "linf_abs_err": float(err.max()), "max_rel_err_where_pi_ge_1_over_n": rel,
            "guarantee": f"rel err <= eps = {args.epsilon} for pi >= 1/n (algo.h:455-463)", "holds": bool(rel <= args.epsilon),
            "exact": f"power iteration, 100 iterations (query.h:1192-1224): first {n_cpu} sources on the CPU in f64 "
                     f"(oracle), the others on the GPU",
            "cpu_vs_gpu_exact_linf": cpu_vs_gpu_exact}


def plumbing_only(args):
    """Same launcher / sharding / timing / gather code path as the real run, with gloo on CPU and a stand-in step."""
    import numpy as np
    import torch.distributed as dist
    from fora_amd import synth
    from fora_amd.dist import env_world, max_over_ranks, min_over_ranks, sum_over_ranks, gather_topk
    rank, _, world = env_world()
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if use_dist:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
    n = 1000
    all_sources, mine = my_sources(np, synth, n, args, rank, world)
    k = max(1, args.topk)
    g_ids = None

    def step():
        nonlocal g_ids
        time.sleep(0.01 * (rank + 1))
        ids = np.repeat(mine[:, None], k, axis=1).astype(np.int32)  # stand-in lists: query i -> k copies of its source id
        sc = np.tile(np.arange(k, 0, -1, dtype=np.float64), (len(mine), 1))
        g_ids, _ = gather_topk(ids, sc, len(all_sources), rank, world if use_dist else 1)
    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if use_dist:
        dist.barrier()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, world if use_dist else 1)
    tot = sum_over_ranks([len(mine)], world if use_dist else 1)
    my_qps = len(mine) * args.steps / dt_local
    ranks = {"world_size_seen": int(dist.get_world_size()) if use_dist else 1, "backend": dist.get_backend() if use_dist else None,
             "rccl": bool(use_dist and dist.get_backend() == "nccl"),
             "per_rank_queries_per_s": {"min": min_over_ranks(my_qps, world if use_dist else 1),
                                        "max": max_over_ranks(my_qps, world if use_dist else 1)}}
    ok = bool((g_ids[:, 0] == all_sources).all())
    per_rank = {"min": int(min_over_ranks(len(mine), world if use_dist else 1)), "max": int(max_over_ranks(len(mine), world if use_dist else 1))}  # (collectives: every rank)
    if rank == 0:
        print(json.dumps({"metric": "plumbing-only", "value": tot[0] * args.steps / dt, "unit": "queries/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "higher_is_better": True, "scaling": args.scaling, "queries_total_per_step": int(tot[0]),
                          "queries_per_rank": per_rank,
                          # what the all-gather moves per step: every rank's padded [ceil(Q / G), k] (int32 id, f64 score) lists
                          "gather_bytes_per_step": int(world * ((len(all_sources) + world - 1) // world) * k * 12),
                          "gather_in_global_order": ok, "ranks": ranks}))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def fifo_ratio_file(graph):
    """FIFO / GPU-schedule work ratios of a graph measured once in a builder run (profiles/fifo_counts_<graph>.json,
    tools/fifo_counts.py): one sequential FIFO push of a Twitter-2010-sized source is about a minute of one core, too
    long for the default bench run.  Returns (pop ratio, relaxation ratio, note) or None."""
    path = os.path.join(ROOT, "profiles", f"fifo_counts_{graph}.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return d["fifo_pops"] / d["gpu_pops"], d["fifo_relax"] / d["gpu_relax"], d.get("note", path)


def scale_fifo_counts(fifo_pops, fifo_relax, k, stats, tm, q_timed):
    """Per-query FIFO counts for the WHOLE query set from a CPU sample of its first k sources: sources differ a lot in
    work, so the sample is scaled by the GPU schedule's own per-query counters -- (FIFO / GPU over the sampled sources) x
    (GPU mean over all timed queries).  Returns (pops, relaxations, description)."""
    gp = sum(int(s["pops"]) for s in stats[:k])
    ge = sum(int(s["relax"]) for s in stats[:k])
    if k <= 0 or gp == 0 or ge == 0:
        return None, None, "GPU schedule (no CPU sample)"
    rp, re = fifo_pops / gp, fifo_relax / ge
    return (rp * tm["pops"] / max(1, q_timed), re * tm["relax"] / max(1, q_timed),
            f"sequential FIFO oracle (CPU) on the first {k} sources, scaled to all sources by the GPU schedule's own per-query "
            f"counts (FIFO / GPU: pops {rp:.3f}, relaxations {re:.3f})")


def run_workload(args, ctx, light=False):
    """One workload: graph, engine, warm-up, timed steps, the result object (rank 0; None on the others).
    light: an extra configuration inside the default run -- no CPU legs, no accuracy pass, no variants."""
    torch, dist, np = ctx["torch"], ctx["dist"], ctx["np"]
    import fora_amd
    from fora_amd import synth
    from fora_amd.dist import max_over_ranks, min_over_ranks, sum_over_ranks, gather_topk
    rank, local_rank, world, use_dist, dev = ctx["rank"], ctx["local_rank"], ctx["world"], ctx["use_dist"], ctx["dev"]

    cache = ctx.setdefault("cache", {})
    key = (args.graph, args.dangling)
    t_graph = t_upload = 0.0
    if key in cache:
        n, m, row_ptr, col, eng = cache[key]
    else:
        for k_old in list(cache):  # one graph at a time in HBM and in host memory
            cache.pop(k_old)[4].close()
        t0 = time.perf_counter()
        n, m, row_ptr, col = synth.preset(args.graph, args.dangling)
        t_graph = time.perf_counter() - t0
        eng = fora_amd.Engine(local_rank)
        t0 = time.perf_counter()
        eng.set_graph(n, m, row_ptr, col)
        t_upload = time.perf_counter() - t0
        cache[key] = (n, m, row_ptr, col, eng)
    arch, cus, hbm = eng.device_info()
    eng.set_params(alpha=0.2, epsilon=args.epsilon, opt=args.opt or bool(args.topk), seed=0x464F5241)
    rmax, omega = eng.get_params()
    eng.set_batch(args.batch)
    if args.balanced:
        eng.set_balanced(True, start_scale=args.balanced_start)
    t_idx = 0.0
    if args.with_idx:
        t0 = time.perf_counter()
        eng.build_index()
        t_idx = time.perf_counter() - t0
    else:
        eng.clear_index()

    all_sources, mine = my_sources(np, synth, n, args, rank, world)
    q_step_total = len(all_sources)
    topk_out = {}

    def step():
        if args.topk:
            ids, sc, rounds = eng.topk(mine, args.topk, epsilon=args.epsilon, with_idx=args.with_idx)
            # the one collective of the design: fixed-size lists to every rank, in global query order
            g_ids, g_sc = gather_topk(ids, sc, q_step_total, rank, world if use_dist else 1, dev if use_dist else None)
            topk_out["ids"], topk_out["sc"], topk_out["rounds"] = g_ids, g_sc, rounds
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
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, world if use_dist else 1, dev)
    tm = eng.timing()
    shard = f"sources i mod {world}"
    qdesc = (f"query_size={args.queries}/GPU" if args.scaling == "weak" else f"query_size={args.queries} in total ({shard})")
    # evidence of the N > 1 run in the line itself: what torch.distributed saw, and the spread over the ranks
    my_qps = len(mine) * args.steps / dt_local
    ranks = {"world_size_seen": int(dist.get_world_size()) if use_dist else 1,
             "backend": (dist.get_backend() if use_dist else None),
             "rccl": bool(use_dist and dist.get_backend() == "nccl"),
             "per_rank_queries_per_s": {"min": min_over_ranks(my_qps, world if use_dist else 1, dev),
                                        "max": max_over_ranks(my_qps, world if use_dist else 1, dev)},
             "timing": "barrier + torch.cuda.synchronize() on both sides of the K steps, MAX over ranks"}
    setup = {"graph": t_graph, "upload": t_upload, "index_build": t_idx}

    if args.topk:
        g_sc = topk_out["sc"]
        assert g_sc.shape == (q_step_total, args.topk)
        assert (np.diff(g_sc, axis=1) <= 0).all()
        rounds_all = sum_over_ranks([float(np.sum(topk_out["rounds"])), float(len(mine))], world if use_dist else 1, dev)
        if rank != 0:
            return None
        topk_roofline = None
        push_ms = tm["push_expand_ms"] + tm["push_accum_ms"] + tm["push_tail_ms"] + tm.get("push_team_ms", 0.0) + tm["push_pop_ms"]
        if not args.no_cpu and push_ms > 0:
            # The pushes of the driver's rounds against 52 B per pop + 24 B per relaxation of the sequential FIFO
            # (algo.h:1020-1093), counted by the oracle for the first sources with the round counts the GPU run took
            # (the push does not depend on the walks, only the number of rounds does) and scaled to all sources by the
            # GPU schedule's own counters of the same sources (one extra un-timed call).
            import oracle_lib as O
            g = O.Graph(n, m, row_ptr, col)
            t1 = time.perf_counter()
            fp = fe = nd = 0
            for i, s_ in enumerate(mine[:32]):
                a_, b_ = O.topk_push_counts(g, int(s_), args.topk, args.epsilon, int(topk_out["rounds"][i]))
                fp += a_; fe += b_; nd += 1
                if time.perf_counter() - t1 > min(args.fifo_sample_seconds, 12.0):
                    break
            t_fifo = time.perf_counter() - t1
            eng.reset_timing()
            _, _, r2 = eng.topk(mine[:nd], args.topk, epsilon=args.epsilon, with_idx=args.with_idx)
            tms = eng.timing()
            if tms["pops"] and tms["relax"] and (np.asarray(r2) == np.asarray(topk_out["rounds"][:nd])).all():
                rp, re = fp / tms["pops"], fe / tms["relax"]
                by = 52.0 * rp * tm["pops"] + 24.0 * re * tm["relax"]
                topk_roofline = {
                    "bound": "hbm", "achieved": by / (push_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": by / (push_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "the pushes of the top-k rounds: fora::k_pushq_bin + fora::k_accum<false> levels and fora::k_push_tail",
                    "push_ms_per_step": push_ms / args.steps,
                    "algorithmic_bytes": "52 B per pop + 24 B per edge relaxation",
                    "algorithmic_counts": f"sequential FIFO top-k pushes (oracle, no walks) of the first {nd} sources with the GPU run's round "
                                          f"counts, scaled to all sources by the GPU schedule's own counts (FIFO / GPU: pops {rp:.3f}, "
                                          f"relaxations {re:.3f}; {t_fifo:.0f} s of one core)",
                    "fifo_relaxations_per_query": re * tm["relax"] / max(1, len(mine) * args.steps),
                    "gpu_relaxations_per_query": tm["relax"] / max(1, len(mine) * args.steps)}
        return {
            "roofline": topk_roofline,
            "metric": "SSPPR top-k queries/sec at eps=0.5", "value": q_step_total * args.steps / dt,
            "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"{args.graph}-sized R-MAT (n={n}, m={m}) topk k={args.topk} --opt"
                                   f"{' --with_idx' if args.with_idx else ''} eps={args.epsilon} {qdesc} on {world}x MI355X, "
                                   f"RCCL all-gather of the [Q, k] lists inside the timed region",
                       "graph": args.graph, "n": n, "m": m, "k": args.topk, "avg_rounds": rounds_all[0] / max(1.0, rounds_all[1]),
                       "queries_per_step": int(q_step_total),
                       "batch": eng.get_batch(), "sharding": shard, "device": arch, "cus": cus,
                       "gather_bytes_per_step": int(q_step_total * args.topk * 12)},
            "ranks": ranks,
            "phases": {k: v for k, v in tm.items() if k.endswith("_ms")},
            "setup_s": setup}
    # sanity inside the bench: every query conserved mass exactly, none was skipped
    assert len(last) == len(mine)
    assert all(s["ppr_sum_fix"] == 1 << 62 for s in last), "mass not conserved"
    nd = sum(1 for s in last if not s["dangling_source"])
    tot = sum_over_ranks([len(mine), nd, tm["walks"], tm["walk_steps"], tm["relax"], tm["pops"]],
                         world if use_dist else 1, dev)
    if rank != 0:
        return None
    qps = tot[0] * args.steps / dt
    out = {
        "metric": "SSPPR queries/sec at eps=0.5", "value": qps, "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {
            "workload": f"{args.graph}-sized R-MAT (n={n}, m={m}, dangling={args.dangling}) eps={args.epsilon} "
                        f"{qdesc}, fora push + "
                        f"{'indexed' if args.with_idx else 'online Philox'} walks"
                        f"{' --opt' if args.opt else ''}{' --balanced' if args.balanced else ''} on {world}x MI355X",
            "graph": args.graph, "n": n, "m": m, "epsilon": args.epsilon, "queries_per_step": int(tot[0]),
            "with_idx": bool(args.with_idx), "opt": bool(args.opt), "balanced": bool(args.balanced), "batch": eng.get_batch(),
            "sharding": shard, "non_dangling_sources": int(tot[1]),
            "rmax": rmax, "omega": omega, "device": arch, "cus": cus,
        },
        "ranks": ranks,
    }
    q_timed = len(mine) * args.steps  # queries this rank ran in the timed region
    p_fifo = e_fifo = None
    counts_from = "GPU schedule (no CPU sample)"
    g = None
    if not args.no_cpu:
        import oracle_lib as O
        per_query_guess = 0.05e-6 * m  # one oracle query: LJ-sized 3.4 s, Twitter-sized about 75 s of one core
        if not light and world == 1 and per_query_guess <= args.cpu_seconds:  # the CPU baseline is timed at N = 1 only
            g = O.Graph(n, m, row_ptr, col)
            index = None
            if args.with_idx:
                rw, off, cnt = eng.get_index()
                index = (rw, off, cnt)
            cpu, fp, fe, fn_ = cpu_baseline(g, mine, rmax, omega, args, index)
            p_fifo, e_fifo, counts_from = scale_fifo_counts(fp, fe, fn_, last, tm, q_timed)
            out["cpu_baseline"] = cpu
            host = host_cpus()
            threads = host["threads_usable"] if args.cpu_threads < 0 else min(args.cpu_threads, host["hardware_threads_allowed"])
            if threads > 1 and cpu["value"] * args.cpu_seconds >= 10:  # every thread runs at least one whole query: only when one fits the budget ten times over
                out["cpu_baseline_all_cores"] = cpu_all_cores(g, mine, rmax, omega, args, index, threads, host, cpu["value"])
        elif per_query_guess <= 5.0:  # N > 1 or an extra configuration: only the algorithmic pop / relaxation counts of the FIFO oracle (push only, a few seconds)
            g = O.Graph(n, m, row_ptr, col)
            t1 = time.perf_counter()
            pp = pr = pn = 0
            for s in mine[:64]:
                ps = O.push_fifo(g, int(s), rmax)
                pp += ps["pops"]; pr += ps["relax"]; pn += 1
                if time.perf_counter() - t1 > 5.0:
                    break
            p_fifo, e_fifo, counts_from = scale_fifo_counts(pp, pr, pn, last, tm, q_timed)
            if world > 1:
                out["cpu_baseline"] = None
                out["cpu_baseline_note"] = "the CPU port is timed at N = 1 only (default `python bench.py`)"
        else:
            fr = fifo_ratio_file(args.graph)
            if fr:
                p_fifo = fr[0] * tm["pops"] / max(1, q_timed)
                e_fifo = fr[1] * tm["relax"] / max(1, q_timed)
                counts_from = f"GPU schedule counts x FIFO/GPU ratios (pops {fr[0]:.3f}, relaxations {fr[1]:.3f}) of a builder run: {fr[2]}"
                out["_fifo_ratio_file"] = fr[:2]
            if args.fifo_sample > 0:  # ... and ONE source through the FIFO oracle here (about a minute of one core at Twitter-2010 size): the in-run fraction
                import oracle_lib as O
                g = O.Graph(n, m, row_ptr, col)
                pp = pr = pn = 0
                t1 = time.perf_counter()
                for s in mine:  # sources differ a lot in work (the first one of the bench list is a near-empty push): keep going for the budget
                    ps = O.push_fifo(g, int(s), rmax)
                    pp += ps["pops"]; pr += ps["relax"]; pn += 1
                    if pn >= args.fifo_sample and time.perf_counter() - t1 > args.fifo_sample_seconds:
                        break
                sp, se, sdesc = scale_fifo_counts(pp, pr, pn, last, tm, q_timed)
                out["_fifo_in_run"] = (sp, se, sdesc + f" ({time.perf_counter() - t1:.0f} s of one core)")
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = (f"one oracle query on this graph needs about {per_query_guess:.0f} s of one core, more than "
                                        f"--cpu-seconds {args.cpu_seconds:g}; raise it to time the CPU port here")
    if not light and not args.no_accuracy and not args.opt:
        if g is None and not args.no_cpu and m <= 200_000_000:
            import oracle_lib as O
            g = O.Graph(n, m, row_ptr, col)
        out["accuracy"] = accuracy(eng, g, mine, n, m, args, np)
    if not light and world == 1 and not args.balanced and not args.no_variants and args.graph in ("webstanford", "small", "tiny"):
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
    if tm["push_expand_launches"] or tm.get("push_team_launches", 0):
        e_unit = e_fifo if e_fifo is not None else tm["relax"] / max(1, q_timed)
        p_unit = p_fifo if p_fifo is not None else tm["pops"] / max(1, q_timed)
        team = tm.get("push_team_launches", 0) > 0  # k_push_team: ONE launch per batch runs every level (fora_team.h)
        # the unit of `achieved` / `traffic`: one launch of the dominant kernel WITH what belongs to it -- bucketed push: a level
        # launch of the bin kernel + its accumulate (the one k_push_tail of a batch counted as a launch of its own); team
        # push: the ONE k_push_team launch of a batch + the k_push_tail launch that finishes its slots (= the push of a batch)
        launches = (tm["push_team_launches"] + tm["push_expand_launches"]  # (bucketed launches beside team ones: a step fell back after a time-out)
                    if team else tm["push_expand_launches"] + tm["push_tail_launches"])
        bucketed = tm["push_accum_launches"] > 0 or team
        # bucketed push: one level (and bin pass) is the kernel PAIR k_pushq_bin + k_accum<false> (same launch
        # count); the pop is split between them, so the pair carries the whole push: 52 B per pop + 24 B per edge
        # relaxation, credited once against the sum of both kernels' durations
        alg_bytes = (24.0 * e_unit + (52.0 * p_unit if bucketed else 0.0)) * q_timed
        step_ms = tm["push_expand_ms"] + tm["push_accum_ms"] + tm["push_tail_ms"] + tm.get("push_team_ms", 0.0)
        avg_ms = step_ms / launches
        achieved = (alg_bytes / launches) / (avg_ms * 1e-3) / 1e9
        traffic = None
        traffic_upper = None
        traffic_note = None
        tpath = args.traffic or os.path.join(ROOT, "profiles", f"pmc_traffic_{args.graph}{'_idx' if args.with_idx else ''}.json")
        if os.path.exists(tpath):
            # HBM-side bytes per launch from rocprofv3 PMC passes of this same workload (FETCH_SIZE and
            # WRITE_SIZE in separate runs, KiB -> bytes; MI355X guide: FETCH_SIZE may under-report wide
            # coalesced reads by up to 2x on gfx950 -- calibration on known byte counts: profiles/r04_pmc_calibration.txt)
            pmc = json.load(open(tpath))
            prefixes = (["fora::k_push_team", "fora::k_push_tail"] if team else
                        ["fora::k_pushq_bin", "fora::k_accum<false", "fora::k_push_tail"] if bucketed else ["fora::k_push_expand"])
            keys = [k for k in pmc if any(k.startswith(p) for p in prefixes)]
            lead = [k for k in pmc if k.startswith(prefixes[0])]
            if lead and all("FETCH_SIZE_bytes_total" in pmc[k] for k in keys):
                # all push kernels of the profiled batch, per level launch (the unit `achieved` uses): the bin
                # kernel's launches plus the one k_push_tail that finishes the small levels
                n_launch = sum(pmc[k].get("FETCH_SIZE_dispatches", 0) for k in keys
                               if k.startswith(prefixes[0]) or (not team and k.startswith("fora::k_push_tail")))
                traffic = sum(pmc[k].get("FETCH_SIZE_bytes_total", 0) + pmc[k].get("WRITE_SIZE_bytes_total", 0)
                              for k in keys) / max(1, n_launch)
                # calibration (profiles/r04_pmc_calibration.txt): FETCH_SIZE counts 64 B per request and a request of a streamed
                # (>= 128 B, aligned) read moves 128 B, so the true bytes lie between the raw sum and this bound
                traffic_upper = sum(2 * pmc[k].get("FETCH_SIZE_bytes_total", 0) + pmc[k].get("WRITE_SIZE_bytes_total", 0)
                                    for k in keys) / max(1, n_launch)
                traffic_note = pmc.get("_note")
        by_kernel = {}
        if tm["push_expand_launches"]:
            by_kernel["k_pushq_bin" if bucketed else "k_push_expand"] = tm["push_expand_ms"] / max(1, tm["push_expand_launches"])
        if tm["push_accum_launches"]:
            by_kernel["k_accum<false>"] = tm["push_accum_ms"] / max(1, tm["push_accum_launches"])
        if team:
            by_kernel["k_push_team"] = tm["push_team_ms"] / tm["push_team_launches"]
        if tm["push_tail_launches"]:
            by_kernel["k_push_tail"] = tm["push_tail_ms"] / tm["push_tail_launches"]
        if tm["push_pop_launches"]:
            by_kernel["k_push_pop"] = tm["push_pop_ms"] / tm["push_pop_launches"]
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_upper_bound": traffic_upper, "traffic_source": traffic_note,
            "kernel": ("fora::k_push_team (every level of a batch's push in one launch, residue resident in LDS)" if team else
                       "fora::k_pushq_bin + fora::k_accum<false> (one level / bin pass of the push; k_push_tail finishes the small levels)"
                       if bucketed else "fora::k_push_expand"),
            "launches": int(launches), "avg_launch_ms": avg_ms, "avg_ms_by_kernel": by_kernel,
            "launch_unit": ("one k_push_team launch + the k_push_tail launch that finishes its slots = the push of one batch" if team else
                            "one level launch of the bin kernel + its accumulate; a batch's k_push_tail counted as one more launch"),
            "algorithmic_bytes_per_launch": alg_bytes / launches,
            "algorithmic_bytes": "52 B per pop + 24 B per edge relaxation" if bucketed else "24 B per edge relaxation",
            "algorithmic_counts": counts_from,
            "fifo_relaxations_per_query": e_unit, "fifo_pops_per_query": p_unit,
            "gpu_relaxations_per_query": tm["relax"] / max(1, q_timed), "gpu_pops_per_query": tm["pops"] / max(1, q_timed),
            "push_total": {  # all push kernels against 52*P + 24*E
                "achieved": (52.0 * p_unit + 24.0 * e_unit) * q_timed
                            / ((step_ms + tm["push_pop_ms"]) * 1e-3) / 1e9,
                "ms": step_ms + tm["push_pop_ms"]},
        }
    if "roofline" in out and ("_fifo_in_run" in out or "_fifo_ratio_file" in out):
        rf = out["roofline"]
        alt = {}
        step_s = rf["push_total"]["ms"] * 1e-3
        if "_fifo_ratio_file" in out:
            rp, re = out.pop("_fifo_ratio_file")
            by = (52.0 * rp * tm["pops"] + 24.0 * re * tm["relax"])
            alt["ratio_file"] = {"frac": by / step_s / 1e9 / HBM_PEAK_GBS, "counts": "GPU schedule counts x FIFO/GPU ratios of a builder run (profiles/fifo_counts_*.json)"}
        if "_fifo_in_run" in out:
            sp, se, sdesc = out.pop("_fifo_in_run")
            if sp is not None:
                alt["in_run_sample"] = {"frac": (52.0 * sp + 24.0 * se) * q_timed / step_s / 1e9 / HBM_PEAK_GBS, "counts": sdesc}
        rf["frac_by_count_source"] = alt
        if "in_run_sample" in alt:  # counts measured in THIS run are the quoted figure; the builder's ratio file is the cross-check
            fr_ = alt["in_run_sample"]["frac"]
            rf["frac"], rf["achieved"] = fr_, fr_ * HBM_PEAK_GBS
            rf["algorithmic_bytes_per_launch"] = fr_ * HBM_PEAK_GBS * 1e9 * (rf["push_total"]["ms"] * 1e-3) / max(1, rf["launches"])
            rf["algorithmic_counts"] = alt["in_run_sample"]["counts"]
            rf["frac_quoted_from"] = "in_run_sample"
            rf["push_total"]["achieved"] = fr_ * HBM_PEAK_GBS
    out.pop("_fifo_in_run", None); out.pop("_fifo_ratio_file", None)
    walk_bytes = (tm["walk_steps"] * 20.0 + tm["walks"] * 16.0) if not args.with_idx else tm["walks"] * 20.0
    if not args.with_idx and tm["walk_ms"] > 0 and tm["walk_steps"]:
        # The walk kernel is the dominant kernel of the headline step and has no HBM roofline (its working set is
        # L2-resident): its rate of divergent gathers (one per step) against what the chip sustains for dependent
        # gathers from a table of that size with nothing else to do (tools/gather_bench.hip), and its unit counters.
        wb = {"kernel": "fora::k_walk_dg (online Philox walks over the degree-grouped copy: one gather per step)",
              "gathers_per_s": tm["walk_steps"] / (tm["walk_ms"] * 1e-3), "steps": tm["walk_steps"] / max(1, args.steps), "walk_ms_per_step": tm["walk_ms"] / args.steps}
        wpath = os.path.join(ROOT, "profiles", "walk_bound.json")
        if os.path.exists(wpath) and args.graph == "webstanford":
            ref = json.load(open(wpath))
            wb["gather_bench_per_s"] = ref.get("gather_bench_per_s")
            wb["frac_of_gather_bench"] = wb["gathers_per_s"] / ref["gather_bench_per_s"] if ref.get("gather_bench_per_s") else None
            for k_ in ("ta_busy", "valu_busy", "floor_ms_all_gathers_hit_l1", "source"):
                wb[k_] = ref.get(k_)
        out["walk_bound"] = wb
    out["phases"] = {
        "push_pop_ms": tm["push_pop_ms"], "push_expand_ms": tm["push_expand_ms"], "push_accum_ms": tm["push_accum_ms"], "push_tail_ms": tm["push_tail_ms"],
        "push_team_ms": tm.get("push_team_ms", 0.0),
        "walk_alloc_ms": tm["walk_alloc_ms"], "walk_ms": tm["walk_ms"], "walk_accum_ms": tm["walk_accum_ms"], "other_ms": tm["other_ms"],
        "batch_ms": tm["batch_ms"], "levels_launched": tm["levels"],
        "walks": tm["walks"], "walk_steps": tm["walk_steps"],
        "walks_per_s": tm["walks"] / max(1e-9, tm["walk_ms"] * 1e-3),
        "walk_algorithmic_GBps": walk_bytes / max(1e-9, tm["walk_ms"] * 1e-3) / 1e9,
        "index_build_s": t_idx, "graph_s": t_graph, "upload_s": t_upload,
    }
    return out


# The other BASELINE.json configurations, measured briefly inside the default run (N = 1) so that the driver's
# BENCH line carries them: config 3 (LiveJournal-sized --with_idx), config 4's one-GPU share (Twitter-2010-sized
# --with_idx, 125 of the 1000 sources) and config 5's (top-k k=500 --opt --with_idx, 125 sources).
EXTRA_CONFIGS = [
    ("config3_livejournal_with_idx", dict(graph="livejournal", with_idx=True, queries=1000, steps=2, warmup=1)),
    ("config4_twitter2010_with_idx_one_gpu_share", dict(graph="twitter2010", with_idx=True, queries=125, steps=1, warmup=1, fifo_sample=1)),
    ("config5_twitter2010_topk500_with_idx_one_gpu_share", dict(graph="twitter2010", with_idx=True, topk=500, queries=125, steps=2, warmup=1)),
]


def summarize(d):
    """compact entry of an extra configuration"""
    e = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
         "steps": d["steps"], "warmup": d["warmup"], "queries_per_step": d["config"].get("queries_per_step"),
         "batch": d["config"].get("batch"), "setup_s": d.get("setup_s") or
         {"graph": d["phases"]["graph_s"], "upload": d["phases"]["upload_s"], "index_build": d["phases"]["index_build_s"]}}
    if d.get("roofline") and "launches" in d["roofline"]:
        r = d["roofline"]
        e["roofline"] = {k: r[k] for k in ("frac", "achieved", "peak", "unit", "kernel", "launches", "avg_launch_ms", "avg_ms_by_kernel",
                                           "algorithmic_bytes", "algorithmic_counts", "fifo_relaxations_per_query",
                                           "gpu_relaxations_per_query", "traffic", "traffic_upper_bound")}
        if "frac_by_count_source" in r:
            e["roofline"]["frac_by_count_source"] = r["frac_by_count_source"]
    elif d.get("roofline"):
        e["roofline"] = d["roofline"]
    ph = d.get("phases", {})
    e["phases_ms_per_step"] = {k: v / max(1, d["steps"]) for k, v in ph.items() if k.endswith("_ms")}
    if d["config"].get("with_idx") and ph.get("walk_ms"):
        e["walk_idx"] = {"kernel": "fora::k_walk_idx", "algorithmic_GBps": ph["walk_algorithmic_GBps"], "algorithmic_bytes": "20 B per indexed walk",
                         "walks_per_s": ph["walks_per_s"], "frac_of_hbm_peak": ph["walk_algorithmic_GBps"] / HBM_PEAK_GBS}
    if "avg_rounds" in d["config"]:
        e["avg_rounds"] = d["config"]["avg_rounds"]
        e["k"] = d["config"]["k"]
    return e


def main():
    args = parse()
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # started by torch.distributed.run
    if args.gpus > 1 and not under_launcher:
        return launch(args)
    if under_launcher and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}", file=sys.stderr)
        return 2
    if args.plumbing_only:
        return plumbing_only(args)
    import torch  # first: the process must use ONE HIP runtime (torch's), the library binds to it
    import torch.distributed as dist
    import numpy as np
    from fora_amd.dist import env_world

    rank, local_rank, world = env_world()
    use_dist = under_launcher
    torch.cuda.set_device(local_rank)
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus
    ctx = {"torch": torch, "dist": dist, "np": np, "rank": rank, "local_rank": local_rank, "world": world,
           "use_dist": use_dist, "dev": torch.device("cuda", local_rank)}
    out = run_workload(args, ctx)
    # the default invocation (the driver's BENCH line) also measures the other BASELINE configurations, briefly
    default_run = (world == 1 and args.graph == "webstanford" and args.dangling == "none" and not args.with_idx and not args.topk
                   and not args.opt and not args.balanced and not args.batch and args.epsilon == 0.5 and not args.no_configs)
    if default_run and out is not None:
        import copy
        t_extra = time.perf_counter()
        if not args.no_variants:
            # the plain R-MAT graph (about 43 % of the nodes have no out-edge; dangling sources finish at once,
            # dangling targets send their mass back to the source, algo.h:993-999): q/s over all and over the non-dangling sources
            a = copy.copy(args); a.dangling = "rmat"; a.steps = 3; a.warmup = 1  # (light: no CPU timing legs, but the FIFO push counts of its first sources)
            try:
                d_all = run_workload(a, ctx, light=True)
                n_nd = d_all["config"]["non_dangling_sources"]
                eng = ctx["cache"][("webstanford", "rmat")][4]
                from fora_amd import synth
                srcs = synth.query_set(d_all["config"]["n"], a.queries, 20261001)
                _, st = eng.query(srcs, want_ppr=False)
                nds = srcs[[i for i, x in enumerate(st) if not x["dangling_source"]]]
                eng.query(nds, want_ppr=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    eng.query(nds, want_ppr=False)
                torch.cuda.synchronize()
                dt_nd = time.perf_counter() - t0
                out.setdefault("variants", {})["dangling_rmat"] = {
                    "graph": d_all["config"]["workload"], "value_all_sources": d_all["value"],
                    "value_non_dangling_sources": len(nds) * a.steps / dt_nd, "unit": "queries/s",
                    "sources": int(a.queries), "non_dangling_sources": int(n_nd), "steps": a.steps,
                    "roofline_frac": d_all.get("roofline", {}).get("frac"),  # on FIFO counts like the headline (round 5: GPU counts)
                    "roofline_counts": d_all.get("roofline", {}).get("algorithmic_counts"),
                    "roofline_kernel": d_all.get("roofline", {}).get("kernel"),
                    "push_ms_by_kernel": d_all.get("roofline", {}).get("avg_ms_by_kernel"),
                    "phases_ms_per_step": {k: v / a.steps for k, v in d_all["phases"].items() if k.endswith("_ms")}}
            except Exception as e:  # never lose the headline line to an extra
                out.setdefault("variants", {})["dangling_rmat"] = {"error": repr(e)[:300]}
        out["configs"] = {}
        for name, over in EXTRA_CONFIGS:
            a = copy.copy(args)
            a.no_variants = True; a.no_accuracy = True
            for k, v in over.items():
                setattr(a, k, v)
            try:
                out["configs"][name] = summarize(run_workload(a, ctx, light=True))
            except Exception as e:
                out["configs"][name] = {"error": repr(e)[:300]}
        out["configs"]["_note"] = ("BASELINE.json configs 3-5 measured inside this same run on 1 GPU, few steps each (the headline keys above are "
                                   f"config 2); {time.perf_counter() - t_extra:.0f} s for all extras")
    for k_old in list(ctx.get("cache", {})):
        ctx["cache"].pop(k_old)[4].close()
    if rank == 0 and out is not None:
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
