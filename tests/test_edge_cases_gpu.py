"""Edge cases of the hot path on hand-made graphs, GPU vs twin bit for bit: empty batches, graphs without edges,
hubs far above the per-item limits (PUSH of one node = thousands of relaxations, WALK_SEG = 1024 walks per item),
duplicate edges (the reference keeps them, graph.h:151-161), repeated sources, extreme epsilon, extreme k."""
import numpy as np
import pytest

from conftest import pick_sources

pytestmark = pytest.mark.gpu
SEED = 0x464F5241


def _load(engine, g, **kw):
    engine.clear_index()
    engine.set_graph(g.n, g.m, g.row_ptr, g.col)
    engine.set_params(seed=SEED, **kw)
    return engine.get_params()


def _check_queries(engine, oracle, g, srcs, rmax, omega, **kw):
    ppr, res, st = engine.query_fix(np.asarray(srcs, dtype=np.int32))
    for i, s in enumerate(srcs):
        want, wres, wst = oracle.twin_query(g, int(s), rmax, omega, seed=SEED, **kw)
        assert (res[i] == wres).all() and (ppr[i] == want).all()
        assert st[i]["ppr_sum_fix"] == oracle.FIX_ONE and st[i]["n_walks"] == wst["n_walks"]
    return ppr


def test_empty_batch_and_edgeless_graph(engine, oracle):
    n = 10
    g = oracle.Graph.from_edges(n, 1, np.zeros(0, np.int32), np.zeros(0, np.int32))   # no edge at all: every node dangling
    _load(engine, g, epsilon=0.5)
    ppr, st = engine.query(np.zeros(0, dtype=np.int32))
    assert ppr.shape == (0, n) and len(st) == 0
    ppr, st = engine.query(np.array([3, 7], dtype=np.int32))
    assert (ppr == np.eye(n)[[3, 7]]).all() and all(s["dangling_source"] == 1 and s["n_walks"] == 0 for s in st)  # algo.h:961-965
    _, _, ids, sc = engine.power_iteration(np.array([4], dtype=np.int32), max_iter=5, k=3, want_ppr=False)
    assert ids[0][0] == 4 and abs(sc[0][0] - (1 - 0.8 ** 5)) < 1e-12 and (sc[0][1:] == 0).all()


def test_star_hub_and_duplicate_edges(engine, oracle):
    """Hub with 6000 out-edges (each leaf points back twice -- duplicate edges count twice, graph.h:158-159), a chain
    hanging off one leaf and a dangling end: thousands of relaxations and tens of thousands of walks per node."""
    L = 6000
    n = L + 4
    src = [0] * L + [v for v in range(1, L + 1) for _ in range(2)] + [1, L + 1, L + 2]
    dst = list(range(1, L + 1)) + [0] * (2 * L) + [L + 1, L + 2, L + 3]
    g = oracle.Graph.from_edges(n, len(src), np.array(src, np.int32), np.array(dst, np.int32))
    assert g.deg[0] == L and g.deg[2] == 2 and g.deg[1] == 3 and g.deg[L + 3] == 0
    for eps in (0.5, 0.05):
        rmax, omega = _load(engine, g, epsilon=eps)
        ppr = _check_queries(engine, oracle, g, [0, 1, 2, L + 1, L + 3, 0], rmax, omega)
        assert (ppr[0] == ppr[5]).all()                                  # a source may repeat inside a batch
    engine.build_index()
    idx = engine.get_index()
    assert idx[2][0] == int(np.ceil(L * rmax * omega))                   # build.h:328: hub gets ceil(outdeg*rmax*omega) walks
    pi, _, st = engine.query_fix(np.array([0, 5], dtype=np.int32), with_idx=True, want_residue=False)
    for i, s in enumerate([0, 5]):
        want, _, _ = oracle.twin_query(g, s, rmax, omega, seed=SEED, index=idx)
        assert (pi[i] == want).all() and st[i]["n_idx_hit"] == st[i]["n_walks"]


@pytest.mark.parametrize("eps", [0.02, 3.0])
def test_extreme_epsilon(engine, oracle, tiny_dangling, eps):
    g = tiny_dangling
    rmax, omega = _load(engine, g, epsilon=eps)
    srcs = list(pick_sources(g, 3, 201)) + list(pick_sources(g, 1, 202, want_dangling=True))
    _check_queries(engine, oracle, g, srcs, rmax, omega)
    _load(engine, g, epsilon=eps, opt=True)
    rmax_o, omega_o = engine.get_params()
    _check_queries(engine, oracle, g, srcs[:2], rmax_o, omega_o, opt=True)


def test_extreme_k(engine, oracle, tiny):
    g = tiny
    _load(engine, g, epsilon=0.5, opt=True)
    srcs = pick_sources(g, 3, 203)
    # k > 1 (query.h:1318); 1024 = this build's cap; from k > n/10 the --opt driver's first delta 1/(10k) is below 1/n
    # and the reference's round loop (query.h:1001) never runs: all-zero lists, zero rounds
    for k in (2, 150, 1024):
        ids, sc, rounds = engine.topk(srcs, k, epsilon=0.5)
        bi, bs, br = engine.topk_bound(srcs, k, epsilon=0.5)
        for i, s in enumerate(srcs):
            wid, wsc, wr, _ = oracle.twin_topk_query(g, int(s), k, 0.5, seed=SEED)
            assert rounds[i] == wr and (ids[i] == wid).all() and (sc[i] == wsc).all()
            assert (wr == 0) == (k > g.n // 10)
            wid, wsc, wr, _, _, _ = oracle.twin_topk_bound_query(g, int(s), k, 0.5, seed=SEED)
            assert br[i] == wr and (bi[i] == wid).all() and (bs[i] == wsc).all()


def test_bucket_overflow_is_retried_with_larger_buckets(engine, oracle, small, monkeypatch):
    """Buckets and overflow list far too small for the graph: the call doubles the bucket capacity and runs again until
    the level fits (8 -> 65 536 messages here) instead of failing; results are the twin's bits as always."""
    g = small
    engine.set_option("bkcap", 8)
    engine.set_option("ovcap", 64)
    engine.set_option("tail", 0)
    rmax, omega = _load(engine, g, epsilon=0.5)
    srcs = pick_sources(g, 4, 301)
    ppr, res, st = engine.query_fix(srcs)
    for i, s in enumerate(srcs):
        want, wres, _ = oracle.twin_query(g, int(s), rmax, omega, seed=SEED)
        assert (res[i] == wres).all() and (ppr[i] == want).all()
    ids, sc, rounds = engine.topk_bound(srcs[:2], 50, epsilon=0.5)   # the enlarged plan is kept for later calls
    wid, wsc, wr, _, _, _ = oracle.twin_topk_bound_query(g, int(srcs[0]), 50, 0.5, seed=SEED)
    assert rounds[0] == wr and (ids[0] == wid).all() and (sc[0] == wsc).all()
    engine.reset_options()
    engine.set_graph(g.n, g.m, g.row_ptr, g.col)   # a new graph starts from the default capacity again


def test_topk_small_buckets_overflow_is_retried(engine, oracle, small):
    """The top-k driver plans its message buckets at 1 / topk_bk_div of a query's on graphs of the wide layout (round 5: 37
    instead of 8 Twitter-2010-sized slots per batch).  With a divisor far too large the buckets (and the overflow list)
    overflow in the later rounds: the call doubles them and runs again by itself; ids, scores and rounds are the twin's."""
    g = small
    engine.set_option("force_wide", 1)
    engine.set_option("tail", 0)          # every level through the bucketed kernels
    engine.set_option("ovcap", 1024)      # (the wide layout sends increments >= 2^50 through this list: not much smaller)
    engine.set_option("xb", 1)            # one sub-bucket per (slot, bin)
    engine.set_option("topk_bk_div", 1 << 10)
    try:
        engine.clear_index()
        engine.set_graph(g.n, g.m, g.row_ptr, g.col)
        engine.set_params(epsilon=0.5, opt=True, seed=SEED)
        srcs = pick_sources(g, 3, 311)
        r0 = engine.get_option("bucket_retries")
        ids, sc, rounds = engine.topk(srcs, 100, epsilon=0.5)
        assert engine.get_option("bucket_retries") > r0   # the buckets did overflow and the call was run again
        for i, s in enumerate(srcs):
            wid, wsc, wr, _ = oracle.twin_topk_query(g, int(s), 100, 0.5, seed=SEED)
            assert rounds[i] == wr and (ids[i] == wid).all() and (sc[i] == wsc).all()
        # the default divisor on the same (forced wide) layout: same bits, no retry needed
        engine.set_option("topk_bk_div", 16)
        engine.set_option("xb", 0)
        engine.set_option("ovcap", 0)
        engine.set_graph(g.n, g.m, g.row_ptr, g.col)  # (back to the default capacity)
        engine.set_params(epsilon=0.5, opt=True, seed=SEED)
        r1 = engine.get_option("bucket_retries")
        ids2, sc2, rounds2 = engine.topk(srcs, 100, epsilon=0.5)
        assert engine.get_option("bucket_retries") == r1
        assert (ids2 == ids).all() and (sc2 == sc).all() and (rounds2 == rounds).all()
    finally:
        engine.reset_options()
        engine.set_graph(g.n, g.m, g.row_ptr, g.col)


def test_c_binding_runs_end_to_end():
    """INTEGRATION.md's call sequence, compiled as C and linked against libfora_hip.so, on the GPU."""
    import os
    import subprocess
    from fora_amd import build
    exe = build.build_c_smoke()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c smoke ok" in r.stdout


@pytest.mark.parametrize("seed", range(12))
def test_random_small_graphs_vs_twin(engine, oracle, seed):
    """Differential check on a dozen random graph shapes -- sparse / dense, with and without dangling nodes, power-law and
    uniform degrees, duplicate edges, fewer nodes than hub records, sizes just around the 64-id blocks and 8192-node bins
    of the internal copies: query (online walks over the degree-grouped copy, hub pre-aggregation from the first level on),
    --opt, an indexed query and top-k equal the twin bit for bit."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    n = int([37, 64, 65, 200, 257, 1000, 4095, 8192, 8193, 9000, 20000, 33000][seed])
    avg = float(rng.choice([0.7, 2.0, 6.0, 20.0]))
    m = max(1, int(n * avg))
    if seed % 3 == 0:   # power-law sources and targets, duplicates kept (graph.h:158)
        src = np.minimum((n * rng.random(m) ** 3).astype(np.int64), n - 1)
        dst = np.minimum((n * rng.random(m) ** 2).astype(np.int64), n - 1)
    elif seed % 3 == 1:  # uniform, many dangling nodes when avg < 1
        src = rng.integers(0, n, m)
        dst = rng.integers(0, n, m)
    else:                # a few giant rows over a sparse background
        hubs = rng.integers(0, n, 3)
        src = np.concatenate([rng.integers(0, n, m // 2), rng.choice(hubs, m - m // 2)])
        dst = rng.integers(0, n, m)
    g = oracle.Graph.from_edges(n, m, src.astype(np.int32), dst.astype(np.int32))
    engine.set_option("hub_min", 1)
    engine.set_option("tail", 64)
    try:
        for opt in (False, True):
            rmax, omega = _load(engine, g, epsilon=0.5, opt=opt)
            srcs = rng.integers(0, n, 5).astype(np.int32)
            _check_queries(engine, oracle, g, srcs, rmax, omega, opt=opt)
        # opt=True parameters are loaded: index + top-k in the driver's flavour
        engine.build_index()
        idx = engine.get_index()
        pi, _, st = engine.query_fix(srcs[:2], with_idx=True, want_residue=False)
        for i in range(2):
            want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, opt=True, seed=SEED, index=idx)
            assert (pi[i] == want).all() and st[i]["n_idx_hit"] == wst["n_idx_hit"]
        k = 8
        if n - 1 > k >= 2:
            ids, sc, rounds = engine.topk(srcs[:2], k, epsilon=0.5, with_idx=True)
            for i in range(2):
                wid, wsc, wr, _ = oracle.twin_topk_query(g, int(srcs[i]), k, 0.5, seed=SEED, index=idx)
                assert rounds[i] == wr and (ids[i] == wid).all() and (sc[i] == wsc).all()
    finally:
        engine.clear_index()
        engine.reset_options()
