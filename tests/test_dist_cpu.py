"""Multi-rank path on CPU: world_size-2 gloo run of the sharding + top-k gather used by
bench.py / multi-GPU drivers.  The per-rank compute is the oracle twin here
(no GPU in this container); on a node each rank runs the HIP engine instead."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
from fora_amd import synth
from fora_amd.dist import env_world, shard_sources, gather_topk, max_over_ranks, sum_over_ranks
import oracle_lib as O
rank, local_rank, world = env_world()
dist.init_process_group("gloo")
n, m, seed = synth.PRESETS["tiny"]
src, dst = synth.rmat_graph(n, m, seed)
g = O.Graph.from_edges(n, m, src, dst)
sources = synth.query_set(n, 7, 99)          # 7 queries over 2 ranks: ragged shards (4 + 3)
mine = shard_sources(sources, rank, world)
k = 8
ids = np.zeros((len(mine), k), dtype=np.int32); sc = np.zeros((len(mine), k))
for i, s in enumerate(mine):
    ids[i], sc[i], _, _ = O.twin_topk_query(g, int(s), k, 0.5, seed=5)
dist.barrier()
all_ids, all_sc = gather_topk(ids, sc, len(sources), rank, world)
t = max_over_ranks(1.0 + rank, world)
tot = sum_over_ranks([len(mine)], world)
if rank == 0:
    ok = True
    for qi, s in enumerate(sources):
        wi, ws, _, _ = O.twin_topk_query(g, int(s), k, 0.5, seed=5)
        ok &= bool((all_ids[qi] == wi).all() and (all_sc[qi] == ws).all())
    print("RESULT", ok, t, tot[0])
dist.destroy_process_group()
'''


def test_world2_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert line[1] == "True" and float(line[2]) == 2.0 and float(line[3]) == 7.0


def test_shard_layout():
    from fora_amd.dist import shard_sources, unshard_index
    src = np.arange(10)
    parts = [shard_sources(src, r, 4) for r in range(4)]
    assert [p.tolist() for p in parts] == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    per = 3
    flat = np.full(4 * per, -1)
    for r, p in enumerate(parts):
        flat[r * per:r * per + len(p)] = p
    assert flat[unshard_index(10, 4)].tolist() == list(range(10))


def _bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-only", "--steps", "2", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_gpus_flag_launches_the_ranks():
    """`bench.py --gpus 2` started plainly launches its two ranks itself (torch.distributed.run, fresh children) and
    rank 0 reports n_gpus = 2; weak scaling runs --queries per rank, strong scaling shards --queries, and the
    top-k gather inside the timed region returns the lists in global query order.  --plumbing-only swaps the GPU step
    for a stand-in and RCCL for gloo; launcher, sharding, barrier / max-over-ranks timing and gather are bench.py's own."""
    weak = _bench("--gpus", "2", "--queries", "7", "--topk", "4")
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["queries_total_per_step"] == 14
    assert weak["gather_in_global_order"] is True
    assert weak["ms_per_step"] >= 20.0  # the slower rank (rank 1 sleeps 20 ms per step) sets the time: max over ranks
    # the line itself shows what torch.distributed saw and the spread over the ranks (rank 0 sleeps 10 ms, rank 1 20 ms)
    assert weak["ranks"]["world_size_seen"] == 2 and weak["ranks"]["backend"] == "gloo" and weak["ranks"]["rccl"] is False
    assert weak["ranks"]["per_rank_queries_per_s"]["min"] < weak["ranks"]["per_rank_queries_per_s"]["max"]
    strong = _bench("--gpus", "2", "--queries", "7", "--topk", "4", "--scaling", "strong")
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["queries_total_per_step"] == 7
    assert strong["gather_in_global_order"] is True
    one = _bench("--gpus", "1", "--queries", "5", "--topk", "3")
    assert one["n_gpus"] == 1 and one["queries_total_per_step"] == 5 and one["ranks"]["world_size_seen"] == 1


def test_bench_eight_ranks_config5_shape():
    """Pre-flight of the driver's 8-GPU run of BASELINE config 5 (Twitter-2010 top-k, k = 500, 1000 sources sharded over the
    node): 8 gloo ranks through bench.py's own launcher -- 125 sources per rank, the gathered lists in global query order,
    and 6 000 000 bytes per step through the all-gather (8 x 125 x 500 x (4 + 8))."""
    line = _bench("--gpus", "8", "--scaling", "strong", "--queries", "1000", "--topk", "500")
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["queries_total_per_step"] == 1000
    assert line["queries_per_rank"] == {"min": 125, "max": 125}
    assert line["gather_in_global_order"] is True and line["gather_bytes_per_step"] == 6_000_000
    assert line["ranks"]["world_size_seen"] == 8


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-only", "--gpus", "4"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_fifo_counts_are_scaled_by_the_gpu_counters():
    """bench.py credits the push with the FIFO oracle's pops / relaxations; its CPU sample covers the first sources only,
    so it is scaled to the whole query set by the GPU schedule's own per-query counters."""
    sys.path.insert(0, ROOT)
    import bench
    stats = [{"pops": 100, "relax": 1000}, {"pops": 300, "relax": 3000}, {"pops": 10, "relax": 50}, {"pops": 90, "relax": 950}]
    tm = {"pops": 2 * 500, "relax": 2 * 5000}  # two timed steps over the four queries
    p, e, how = bench.scale_fifo_counts(fifo_pops=320, fifo_relax=3000, k=2, stats=stats, tm=tm, q_timed=8)
    assert abs(p - 0.8 * 125) < 1e-9 and abs(e - 0.75 * 1250) < 1e-9      # (FIFO / GPU on the first two) x (GPU mean over all)
    assert how.startswith("sequential FIFO oracle") and "first 2 sources" in how
    assert bench.scale_fifo_counts(0, 0, 0, stats, tm, 8)[0] is None
