"""GPU parity tests, all through the C ABI (fora_amd.Engine -> libfora_hip.so).

Bars:
  * vs the oracle TWIN (same schedule, same fixed point): bit-exact on reserve,
    residue, rsum, counters, walk counts, walk endpoints, index contents, final ppr.
  * vs the oracle in the REFERENCE's arithmetic and FIFO order: walk allocation and
    walk endpoints bit-exact on the same inputs; refined ppr within a stated L-inf
    tolerance; both within the epsilon guarantee of exact PPR.
"""
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


@pytest.mark.parametrize("gname", ["tiny", "tiny_dangling", "small", "small_dangling"])
def test_push_bit_exact_vs_twin(engine, oracle, request, gname):
    g = request.getfixturevalue(gname)
    rmax, omega = _load(engine, g, epsilon=0.5)
    assert (rmax, omega) == oracle.fora_setting(g.n, g.m, 0.5)
    srcs = np.concatenate([pick_sources(g, 6, 21), pick_sources(g, 2, 22, want_dangling=True)])
    rsv, res, st = engine.push(srcs)
    for i, s in enumerate(srcs):
        t = oracle.twin_push(g, int(s), rmax)
        assert (rsv[i] == t["reserve"]).all()
        assert (res[i] == t["residue"]).all()
        assert st[i]["rsum_fix"] == t["rsum_fix"]
        assert st[i]["pops"] == t["pops"] and st[i]["relax"] == t["relax"] and st[i]["levels"] == t["levels"]
        assert int(rsv[i].sum()) + int(res[i].sum()) == oracle.FIX_ONE
        assert st[i]["dangling_source"] == int(g.deg[s] == 0)


@pytest.mark.parametrize("opt", [False, True])
def test_walk_allocation_bit_exact_vs_reference_arithmetic(engine, oracle, small, opt):
    """num_s_rw from the FIFO oracle's f64 residues: query.h:282 / :364, same device code
    (fora::walk_count) the product path uses."""
    g = small
    rmax, omega = _load(engine, g, epsilon=0.5, opt=opt)
    assert (rmax, omega) == oracle.fora_setting(g.n, g.m, 0.5, opt=opt)
    for s in pick_sources(g, 3, 23):
        p = oracle.push_fifo(g, int(s), rmax)
        N, cnt = oracle.walk_counts(p, omega, opt=opt)
        dense = np.zeros(g.n, dtype=np.uint64)
        dense[p["residue_occur"]] = cnt
        N2, got = engine.walk_counts(p["residue"], p["rsum"])
        assert N2 == N
        assert (got == dense).all()


def test_walk_endpoints_bit_exact(engine, oracle, small_dangling):
    g = small_dangling
    _load(engine, g, epsilon=0.5)
    rng = np.random.Generator(np.random.PCG64(24))
    starts = rng.integers(0, g.n, size=4000).astype(np.int32)
    js = rng.integers(0, 1 << 44, size=4000).astype(np.uint64)
    for nzh in (False, True):
        for rnd, stream in ((0, 12345), (3, oracle.STREAM_INDEX)):
            got = engine.walks(stream, rnd, starts, js, no_zero_hop=nzh)
            want = np.array([oracle.walk(g, SEED, stream, rnd, int(v), int(j), no_zero_hop=nzh)
                             for v, j in zip(starts, js)], dtype=np.int32)
            assert (got == want).all()


def test_long_walks_bit_exact(engine, oracle, tiny):
    """alpha = 0.004: 13 % of the walks outlive the 512-step period of the counter's step field and must still end
    (the last counter word moves on every 512 steps) at the oracle's endpoints."""
    g = tiny
    engine.clear_index()
    engine.set_graph(g.n, g.m, g.row_ptr, g.col)
    engine.set_params(alpha=0.004, epsilon=0.5, seed=SEED)
    rng = np.random.Generator(np.random.PCG64(31))
    starts = rng.choice(np.flatnonzero(g.deg > 0), size=600).astype(np.int32)
    js = rng.integers(0, 1 << 44, size=600).astype(np.uint64)
    got = engine.walks(9, 0, starts, js)
    want = np.array([oracle.walk(g, SEED, 9, 0, int(v), int(j), alpha=0.004) for v, j in zip(starts, js)], dtype=np.int32)
    assert (got == want).all()
    assert sum(oracle.walk_steps(g, SEED, 9, 0, int(v), int(j), alpha=0.004) > 512 for v, j in zip(starts[:100], js[:100])) >= 3
    # and through the product path: an online-walk query at this alpha conserves mass exactly and matches the twin
    rmax, omega = engine.get_params()
    ppr, _, st = engine.query_fix(starts[:2])
    for i in range(2):
        want, _, _ = oracle.twin_query(g, int(starts[i]), rmax, omega, alpha=0.004, seed=SEED)
        assert (ppr[i] == want).all() and st[i]["ppr_sum_fix"] == oracle.FIX_ONE


@pytest.mark.parametrize("opt", [False, True])
@pytest.mark.parametrize("gname", ["tiny_dangling", "small"])
def test_query_bit_exact_vs_twin(engine, oracle, request, gname, opt):
    g = request.getfixturevalue(gname)
    rmax, omega = _load(engine, g, epsilon=0.5, opt=opt)
    srcs = np.concatenate([pick_sources(g, 4, 25), pick_sources(g, 1, 26, want_dangling=True)])
    ppr, res, st = engine.query_fix(srcs)
    for i, s in enumerate(srcs):
        want, wres, wst = oracle.twin_query(g, int(s), rmax, omega, opt=opt, seed=SEED)
        assert (res[i] == wres).all()
        assert (ppr[i] == want).all()
        assert st[i]["n_walks"] == wst["n_walks"] and st[i]["n_idx_hit"] == 0
        assert st[i]["ppr_sum_fix"] == int(want.sum())
        if wst["rsum_fix"]:
            assert st[i]["ppr_sum_fix"] == oracle.FIX_ONE  # mass conserved exactly


@pytest.mark.parametrize("opt", [False, True])
@pytest.mark.parametrize("copy", ["dg_bucket_order", "dg_gather", "dg_many_hubs", "compact", "csr"])
def test_walk_graph_copies_agree(engine, oracle, small_dangling, copy, opt):
    """The online walks step through one of several copies of the graph, chosen at set_graph: the degree-grouped copy
    (k_walk_dg, the default in the narrow layout: one gather per step; results in bucket order with hub endpoints summed
    in LDS, or with one gather per walk for the endpoint's id; with the fewest or with 1024 hub records), the bit-packed
    compact copy or the plain CSR (k_walk_online).  Same edge choice in all of them: queries -- with and without indexed
    walks in front of the online ones -- and the index build equal the twin bit for bit."""
    g = small_dangling
    engine.set_option("no_compact", 1 if copy == "csr" else 0)
    engine.set_option("walk_dg", {"dg_bucket_order": 2, "dg_many_hubs": 2, "dg_gather": 1}.get(copy, 0))
    if copy == "dg_many_hubs":
        engine.set_option("dg_hubs", 1024)
    try:
        rmax, omega = _load(engine, g, epsilon=0.5, opt=opt)
        srcs = np.concatenate([pick_sources(g, 3, 25), pick_sources(g, 1, 26, want_dangling=True)])
        ppr, _, st = engine.query_fix(srcs, want_residue=False)
        for i, s in enumerate(srcs):
            want, _, wst = oracle.twin_query(g, int(s), rmax, omega, opt=opt, seed=SEED)
            assert (ppr[i] == want).all() and st[i]["n_walks"] == wst["n_walks"]
        engine.build_index()
        rw, off, cnt = engine.get_index()
        want_rw, _, _ = oracle.build_index(g, SEED, rmax, omega, opt=opt)
        assert (rw == want_rw).all()
        # indexed walks first, online walks for what the index does not cover (query.h:290-307, 320-323): an index cut to
        # a third of its entries per node makes both kernels work on the same query
        cnt3 = cnt // 3
        engine.set_index(rw, off, cnt3)
        ppr, _, st = engine.query_fix(srcs[:3], with_idx=True, want_residue=False)
        for i in range(3):
            want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, opt=opt, seed=SEED, index=(rw, off, cnt3))
            assert (ppr[i] == want).all() and st[i]["n_walks"] == wst["n_walks"] and st[i]["n_idx_hit"] == wst["n_idx_hit"]
            assert st[i]["dangling_source"] or 0 < st[i]["n_idx_hit"] < st[i]["n_walks"]
    finally:
        engine.reset_options()
        engine.clear_index()
        engine.set_graph(g.n, g.m, g.row_ptr, g.col)


@pytest.mark.parametrize("hubs,hub_min,wide", [(0, 1, False), (64, 1, False), (1024, 1, False), (1024, 200, False), (6144, 1, False),
                                                  (512, 1, True), (2048, 100, True)])
def test_hub_preaggregation_bit_exact(engine, oracle, small_dangling, hubs, hub_min, wide):
    """Hub pre-aggregation of the narrow push (options "hubs", read by set_graph, and "hub_min"): increments for the
    nodes of largest in-degree are summed per workgroup in LDS and reach the accumulate as one dense row of sums per
    workgroup.  Integer adds commute: push, query and top-k equal the twin bit for bit for any number of hubs and from
    any level size on."""
    g = small_dangling
    if wide:  # the wide layout (8-byte messages) in one pass per level; option "hubs_wide"
        engine.set_option("force_wide", 1)
        engine.set_option("hubs_wide", hubs)
    else:
        engine.set_option("hubs", hubs)
    engine.set_option("hub_min", hub_min)
    engine.set_option("tail", 0)  # every level through the bucketed kernels
    engine.set_option("team", 0)  # ... not through k_push_team
    try:
        rmax, omega = _load(engine, g, epsilon=0.5)
        srcs = np.concatenate([pick_sources(g, 6, 91), pick_sources(g, 1, 92, want_dangling=True)])
        rsv, res, st = engine.push(srcs)
        for i, s in enumerate(srcs):
            t = oracle.twin_push(g, int(s), rmax)
            assert (res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all()
            assert st[i]["pops"] == t["pops"] and st[i]["relax"] == t["relax"] and st[i]["levels"] == t["levels"]
        ppr, _, stq = engine.query_fix(srcs[:3], want_residue=False)
        for i in range(3):
            want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, seed=SEED)
            assert (ppr[i] == want).all() and stq[i]["n_walks"] == wst["n_walks"]
        engine.set_params(epsilon=0.5, opt=True, seed=SEED)
        ids, sc, rounds = engine.topk(srcs[:2], 50, epsilon=0.5)
        for i in range(2):
            wid, wsc, wr, _ = oracle.twin_topk_query(g, int(srcs[i]), 50, 0.5, seed=SEED)
            assert rounds[i] == wr and (ids[i] == wid).all() and (sc[i] == wsc).all()
    finally:
        engine.reset_options()
        engine.set_graph(g.n, g.m, g.row_ptr, g.col)


@pytest.mark.parametrize("opt", [False, True])
def test_index_build_and_indexed_query_bit_exact(engine, oracle, small, opt):
    g = small
    rmax, omega = _load(engine, g, epsilon=0.5, opt=opt)
    total, off, cnt = engine.index_sizes()
    t2, off2, cnt2 = oracle.index_sizes(g, rmax, omega, opt=opt)
    assert total == t2 and (off == off2).all() and (cnt == cnt2).all()   # build.h:325-334
    engine.build_index()
    rw, o3, c3 = engine.get_index()
    want_rw, _, _ = oracle.build_index(g, SEED, rmax, omega, opt=opt)
    assert (rw == want_rw).all()                                        # build.h:344-354 under Philox
    srcs = pick_sources(g, 4, 27)
    ppr, res, st = engine.query_fix(srcs, with_idx=True)
    for i, s in enumerate(srcs):
        want, _, wst = oracle.twin_query(g, int(s), rmax, omega, opt=opt, seed=SEED, index=(want_rw, off2, cnt2))
        assert (ppr[i] == want).all()
        assert st[i]["n_idx_hit"] == wst["n_idx_hit"] == st[i]["n_walks"]  # 100 % index hit
    # an index that is too short forces the online top-up branch (query.h:290-300)
    half = (cnt2 // 2).astype(np.uint64)
    engine.set_index(want_rw, off2, half)
    ppr, _, st = engine.query_fix(srcs, with_idx=True)
    for i, s in enumerate(srcs):
        want, _, wst = oracle.twin_query(g, int(s), rmax, omega, opt=opt, seed=SEED, index=(want_rw, off2, half))
        assert (ppr[i] == want).all()
        assert st[i]["n_idx_hit"] == wst["n_idx_hit"] < st[i]["n_walks"]
    engine.clear_index()


@pytest.mark.parametrize("opt", [False, True])
def test_vs_reference_order_oracle_and_exact_ppr(engine, oracle, small, opt):
    """The float side of the bar: |ppr_gpu - ppr_fifo_oracle|_inf <= 1e-3 (two valid push
    states + independent Monte-Carlo noise at n=32k), and the FORA guarantee
    |est - pi| <= eps*pi wherever pi >= 1/n, against power iteration."""
    g = small
    eps = 0.5
    rmax, omega = _load(engine, g, epsilon=eps, opt=opt)
    srcs = pick_sources(g, 3, 28)
    ppr, st = engine.query(srcs)
    for i, s in enumerate(srcs):
        ref, rst = oracle.query(g, int(s), rmax, omega, opt=opt, seed=SEED)
        exact = oracle.power_iteration(g, int(s))
        assert abs(ppr[i].sum() - 1) < 1e-12
        assert np.abs(ppr[i] - ref).max() <= 1e-3
        big = exact >= 1.0 / g.n
        assert (np.abs(ppr[i] - exact)[big] / exact[big]).max() <= eps
        # reserve side: push states differ by schedule only; either push stops with every residue under rmax * outdeg
        assert 0 <= st[i]["rsum"] < g.col.size * rmax and 0 <= rst["rsum"] < g.col.size * rmax


def test_batching_and_determinism(engine, oracle, small):
    """Results do not depend on batch size, slot position, or run (integer atomics)."""
    g = small
    _load(engine, g, epsilon=0.5)
    srcs = pick_sources(g, 12, 29)
    engine.set_batch(12)
    a, _, _ = engine.query_fix(srcs, want_residue=False)
    engine.set_batch(5)
    b, _, _ = engine.query_fix(srcs[::-1].copy(), want_residue=False)
    engine.set_batch(0)
    assert (a == b[::-1]).all()
    c, _, _ = engine.query_fix(srcs, want_residue=False)
    assert (a == c).all()


def test_error_behaviour(engine, oracle, tiny):
    import fora_amd
    g = tiny
    _load(engine, g, epsilon=0.5)
    with pytest.raises(fora_amd.ForaError):
        engine.query(np.array([g.n], dtype=np.int32))            # id >= n (graph.h:155)
    with pytest.raises(fora_amd.ForaError):
        engine.query(np.array([-1], dtype=np.int32))
    with pytest.raises(fora_amd.ForaError):
        engine.query(np.array([0], dtype=np.int32), with_idx=True)  # no index loaded
    bad = g.col.copy()
    bad[0] = g.n
    with pytest.raises(fora_amd.ForaError):
        engine.set_graph(g.n, g.m, g.row_ptr, bad)
    _load(engine, g, epsilon=0.5)
    out, st = engine.query(np.zeros(0, dtype=np.int32))          # empty batch is fine
    assert out.shape == (0, g.n)


def test_full_size_webstanford_properties(engine, oracle):
    """BASELINE size (n=281 904, m=2 312 497), online walks and --with_idx: exact mass conservation per query, the
    push's exit condition over ALL nodes, determinism, one query bit for bit against the twin -- and the float side
    of the bar at this size: four sources against the reference-order f64 oracle (FIFO push, query.h:841-907) and
    against exact PPR (CPU power iteration, query.h:1192-1224), with the tolerances stated here."""
    from fora_amd import synth
    n, m, row_ptr, col = synth.preset("webstanford")
    engine.clear_index()
    engine.set_graph(n, m, row_ptr, col)
    engine.set_params(epsilon=0.5, seed=SEED)
    rmax, omega = engine.get_params()
    assert (rmax, omega) == oracle.fora_setting(n, m, 0.5)
    srcs = synth.query_set(n, 64, 7)
    ppr, res, st = engine.query_fix(srcs)
    deg = np.diff(row_ptr)
    t1 = np.uint64(int(np.ceil(np.ldexp(rmax, 62))))
    thr = t1 * deg.astype(np.uint64)   # < 2^63: t1 ~ 4.5e11, degrees < 1e5
    thr[deg == 0] = 1
    for i in range(len(srcs)):
        assert st[i]["ppr_sum_fix"] == 1 << 62
        assert int(ppr[i].sum()) == 1 << 62
        assert (res[i] < thr).all()                      # algo.h:1012 for every node
        assert int(res[i].sum()) == st[i]["rsum_fix"]
    ppr2, _, _ = engine.query_fix(srcs[:8], want_residue=False)
    assert (ppr2 == ppr[:8]).all()
    # one query against the twin at full size (about a second of CPU)
    g = oracle.Graph(n, m, row_ptr, col)
    want, _, _ = oracle.twin_query(g, int(srcs[0]), rmax, omega, seed=SEED)
    assert (ppr[0] == want).all()
    # ---- float side.  Tolerances: the estimate keeps |est - pi| <= eps * pi wherever pi >= 1/n (the FORA guarantee,
    # algo.h:455-463; measured worst 0.25 of the allowed 0.5); L-inf against exact PPR <= 2e-5 (measured 5e-6: walk
    # noise ~ sqrt(pi / omega)); against the FIFO-order oracle, itself such an estimate with its own walks, <= 4e-5.
    # Both pushes stop with every residue under rmax * outdeg, so either rsum is below nnz * rmax = 0.227.
    eps = 0.5
    engine.build_index()
    ppr_idx, _, st_idx = engine.query_fix(srcs[:4], with_idx=True, want_residue=False)
    est_idx = oracle.fix_to_double(ppr_idx)
    idx = engine.get_index()
    for i in range(4):
        s = int(srcs[i])
        est = oracle.fix_to_double(ppr[i])
        exact = oracle.power_iteration(g, s)
        fifo, fst = oracle.query(g, s, rmax, omega, seed=SEED)
        fifo_idx, fst_idx = oracle.query(g, s, rmax, omega, seed=SEED, index=idx)
        big = exact >= 1.0 / n
        for e, f, stats, fstats in ((est, fifo, st[i], fst), (est_idx[i], fifo_idx, st_idx[i], fst_idx)):
            assert abs(e.sum() - 1.0) < 1e-12
            assert np.abs(e - exact).max() <= 2e-5
            assert (np.abs(e - exact)[big] / exact[big]).max() <= eps
            assert (np.abs(f - exact)[big] / exact[big]).max() <= eps       # the oracle meets the same guarantee
            assert np.abs(e - f).max() <= 4e-5
            assert 0 < stats["rsum"] < col.size * rmax and 0 < fstats["rsum"] < col.size * rmax
        assert st_idx[i]["n_idx_hit"] == st_idx[i]["n_walks"] == st[i]["n_walks"]      # same push, 100 % index hit
        assert fst_idx["n_idx_hit"] == fst_idx["n_walks"]
    engine.clear_index()


@pytest.mark.parametrize("with_idx", [False, True])
@pytest.mark.parametrize("gname", ["small", "small_dangling"])
def test_topk_bit_exact_vs_twin(engine, oracle, request, gname, with_idx):
    """fora_query_topk_new + topk_ppr (query.h:972-1045, algo.h:592-610), --opt driver."""
    g = request.getfixturevalue(gname)
    k, eps = 50, 0.5
    rmax, omega = _load(engine, g, epsilon=eps, opt=True)
    index = None
    if with_idx:
        engine.build_index()
        index = engine.get_index()
        want_rw, _, _ = oracle.build_index(g, SEED, rmax, omega, opt=True)
        assert (index[0] == want_rw).all()
    srcs = np.concatenate([pick_sources(g, 5, 31), pick_sources(g, 1, 32, want_dangling=True)])
    ids, sc, rounds = engine.topk(srcs, k, epsilon=eps, with_idx=with_idx)
    for i, s in enumerate(srcs):
        wid, wsc, wr, _ = oracle.twin_topk_query(g, int(s), k, eps, seed=SEED, index=index)
        assert rounds[i] == wr
        assert (ids[i] == wid).all()
        assert (sc[i] == wsc).all()
        # against the reference-order oracle: same top of the list, close scores
        rid, rsc, rr, _ = oracle.topk_query(g, int(s), k, eps, seed=SEED, index=index)
        exact = oracle.power_iteration(g, int(s))
        if g.deg[s] > 0:
            truth = set(np.argsort(-exact)[:k].tolist())
            assert len(truth & set(ids[i].tolist())) >= int(0.8 * k)
            assert len(set(rid.tolist()) & set(ids[i].tolist())) >= int(0.7 * k)
            assert abs(sc[i][0] - rsc[0]) / rsc[0] < 0.1
        else:
            assert ids[i][0] == s and sc[i][0] == 1.0 and (sc[i][1:] == 0).all()
    engine.clear_index()


def test_topk_argument_checks(engine, oracle, tiny):
    import fora_amd
    g = tiny
    _load(engine, g, epsilon=0.5, opt=True)
    with pytest.raises(fora_amd.ForaError):
        engine.topk(np.array([1], dtype=np.int32), g.n, epsilon=0.5)   # assert(k < n-1), query.h:1317
    with pytest.raises(fora_amd.ForaError):
        engine.topk(np.array([1], dtype=np.int32), 1, epsilon=0.5)     # assert(k > 1), query.h:1318
    with pytest.raises(fora_amd.ForaError):
        engine.topk(np.array([1], dtype=np.int32), 10, epsilon=0.5, with_idx=True)


@pytest.mark.parametrize("mode", ["direct", "bucketed", "bucketed_overflow", "bucketed_wide", "bucketed_wide_overflow",
                                  "bucketed_wide_multipass", "bucketed_wide_multipass_unsorted",
                                  "bucketed_wide_multipass_nosplit", "tail_early", "tail_early_nohubs", "tail_wide_multipass"])
def test_push_paths_agree_with_twin(engine, oracle, small_dangling, mode, monkeypatch):
    """The two push organisations (one global atomic per edge; LDS-bucketed) and the bucket
    overflow fallback all give the twin's bits (integer adds commute)."""
    g = small_dangling
    if mode == "bucketed_wide_multipass_unsorted":
        # rows in arbitrary (file) order: the engine pushes over its own row-sorted copy, walks keep file order
        from fora_amd import synth
        n, m, seed = synth.PRESETS["small"]
        src, dst = synth.rmat_graph(n, m, seed, "rmat")
        perm = np.random.Generator(np.random.PCG64(99)).permutation(src.size)
        g = oracle.Graph.from_edges(n, m, src[perm], dst[perm])
        assert (np.diff(g.col[g.row_ptr[np.argmax(g.deg)]:g.row_ptr[np.argmax(g.deg) + 1]]) < 0).any()
        mode = "bucketed_wide_multipass"
    if mode == "bucketed_wide_multipass_nosplit":  # every pass scans whole rows and filters by bin range
        engine.set_option("no_split", 1)
        mode = "bucketed_wide_multipass"
    # the bucketed levels are what these modes are about: keep k_push_tail (which takes over once every frontier is
    # small -- on a 32 k-node graph almost at once) out of them, except where it is the subject
    engine.set_option("tail", 100000000 if mode.startswith("tail") else 0)
    engine.set_option("team", 0)  # the team push has its own test (test_team_push_bit_exact)
    if mode.startswith("tail"):
        engine.set_option("tail_always", 1)
    if mode == "tail_early_nohubs":  # k_push_tail with every relaxation as an atomic (default: hub increments summed in LDS)
        engine.set_option("tail_hubs", 0)
    if mode == "tail_wide_multipass":
        mode = "bucketed_wide_multipass"
    if mode == "direct":
        engine.set_option("direct", 1)
    elif mode == "bucketed_overflow":
        engine.set_option("bkcap", 96)
    elif mode == "bucketed_wide":
        engine.set_option("force_wide", 1)
    elif mode == "bucketed_wide_overflow":
        engine.set_option("force_wide", 1)
        engine.set_option("bkcap", 200)
    elif mode == "bucketed_wide_multipass":  # the layout of graphs with more than 1024 bins: several passes per level
        engine.set_option("force_wide", 1)
        engine.set_option("pass_bins", 1)
    rmax, omega = _load(engine, g, epsilon=0.5)
    srcs = np.concatenate([pick_sources(g, 7, 51), pick_sources(g, 1, 52, want_dangling=True)])
    ppr, res, st = engine.query_fix(srcs)
    for i, s in enumerate(srcs):
        want, wres, wst = oracle.twin_query(g, int(s), rmax, omega, seed=SEED)
        assert (res[i] == wres).all() and (ppr[i] == want).all()
        assert st[i]["pops"] == wst["pops"] and st[i]["relax"] == wst["relax"] and st[i]["levels"] == wst["levels"]
    engine.set_option("direct", 0)
    engine.set_option("bkcap", 0)
    if mode == "bucketed_wide_multipass":  # indexed walks and top-k take the per-pass route too
        engine.build_index()
        idx = engine.get_index()
        pi, _, _ = engine.query_fix(srcs[:3], with_idx=True, want_residue=False)
        for i, s in enumerate(srcs[:3]):
            want, _, _ = oracle.twin_query(g, int(s), rmax, omega, seed=SEED, index=idx)
            assert (pi[i] == want).all()
        engine.clear_index()
    engine.reset_options()
    engine.query_fix(srcs[:1])  # back to the default plan for later tests


@pytest.mark.parametrize("group,slot_major", [(1, 1), (3, 0), (8, 1), (16, 0), (0, 1), (0, 0)])
def test_wide_accumulate_bin_groups_bit_exact(engine, oracle, small_dangling, group, slot_major):
    """k_accum of the wide layouts takes `acc_group` consecutive bins of a slot per workgroup (round 6: the workgroup reads their
    counts in one trip and skips the bins with nothing to do -- overflow entries, hub sums, the dangling mass and walk results
    included).  Same bits for any group size and either dispatch order (option slot_major): push, indexed query
    (k_accum<true, true>, k_walk_idx, k_walk_alloc) and top-k against the twin."""
    g = small_dangling
    engine.set_option("force_wide", 1)
    engine.set_option("acc_group", group)
    engine.set_option("slot_major", 15 * slot_major)  # dispatch order of the wide kernels: slot or tile / bin / chunk fastest (Dev::slot_major)
    engine.set_option("tail", 0)   # every level through the bucketed kernels
    engine.set_option("team", 0)
    try:
        rmax, omega = _load(engine, g, epsilon=0.5)
        srcs = np.concatenate([pick_sources(g, 6, 191), pick_sources(g, 1, 192, want_dangling=True)])
        rsv, res, st = engine.push(srcs)
        for i, s in enumerate(srcs):
            t = oracle.twin_push(g, int(s), rmax)
            assert (res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all()
            assert st[i]["pops"] == t["pops"] and st[i]["relax"] == t["relax"] and st[i]["levels"] == t["levels"]
        engine.set_option("bkcap", 200)  # ... with overflow entries in some bins
        rsv2, res2, _ = engine.push(srcs)
        assert (rsv2 == rsv).all() and (res2 == res).all()
        engine.set_option("bkcap", 0)
        engine.build_index()
        idx = engine.get_index()
        ppr, _, stq = engine.query_fix(srcs[:3], with_idx=True, want_residue=False)
        for i in range(3):
            want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, seed=SEED, index=idx)
            assert (ppr[i] == want).all() and stq[i]["n_walks"] == wst["n_walks"]
        engine.clear_index()
        engine.set_params(epsilon=0.5, opt=True, seed=SEED)
        ids, sc, rounds = engine.topk(srcs[:2], 50, epsilon=0.5)
        for i in range(2):
            wid, wsc, wr, _ = oracle.twin_topk_query(g, int(srcs[i]), 50, 0.5, seed=SEED)
            assert rounds[i] == wr and (ids[i] == wid).all() and (sc[i] == wsc).all()
    finally:
        engine.reset_options()
        engine.clear_index()
        engine.set_graph(g.n, g.m, g.row_ptr, g.col)


@pytest.mark.parametrize("size,tail,xcd,tmax,log", [(0, 0, 1, 0, -1), (0, -1, 1, 0, -1), (4, 0, 1, 0, 0), (4, 64, 0, 0, -1), (16, 300, 1, 0, 40),
                                                    (32, 0, 1, 0, -1), (32, 2000, 0, 0, 7), (1, 16, 1, 3, 1000), (8, 1, 1, 1, -1), (2, 0, 2, 0, 100)])
@pytest.mark.parametrize("gname", ["small_dangling", "small"])
def test_team_push_bit_exact(engine, oracle, request, gname, size, tail, xcd, tmax, log):
    """k_push_team (fora_team.h): the residue of a slot stays in the LDS of a team of workgroups for the whole push; levels
    are the twin's.  Team sizes 1 ... 32 (option team_size; 0 = the fewest members whose LDS holds the graph), the
    hand-over to k_push_tail at several frontier sizes (team_tail; 0 = the team runs the push to its end), both placements
    of the members (team_xcd; 2 = with the fences of members on different XCDs), fewer teams than slots (team_max) and
    reserve logs of every size (team_log: none, a few entries -- most pops go to the accumulators --, the default) all
    give the twin's bits."""
    g = request.getfixturevalue(gname)
    engine.set_option("team", 1)
    engine.set_option("team_size", size)
    engine.set_option("team_xcd", xcd)
    engine.set_option("team_max", tmax)
    engine.set_option("team_log", log)
    if tail >= 0:
        engine.set_option("team_tail", tail)
        engine.set_option("tail_always", 1)  # hand over as soon as the frontier is that small, also while it is still growing
    try:
        rmax, omega = _load(engine, g, epsilon=0.5)
        srcs = np.concatenate([pick_sources(g, 9, 151), pick_sources(g, 2, 152, want_dangling=True)])
        engine.reset_timing()
        rsv, res, st = engine.push(srcs)
        tm = engine.timing()
        assert tm["push_team_launches"] >= 1 and tm["push_expand_launches"] == 0  # the team kernel ran, the bucketed ones did not
        for i, s in enumerate(srcs):
            t = oracle.twin_push(g, int(s), rmax)
            assert (res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all()
            assert st[i]["pops"] == t["pops"] and st[i]["relax"] == t["relax"] and st[i]["levels"] == t["levels"]
            assert st[i]["rsum_fix"] == t["rsum_fix"]
        ppr, _, stq = engine.query_fix(srcs[:4], want_residue=False)
        for i in range(4):
            want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, seed=SEED)
            assert (ppr[i] == want).all() and stq[i]["n_walks"] == wst["n_walks"]
    finally:
        engine.reset_options()
        engine.query_fix(srcs[:1])


@pytest.mark.parametrize("rounds,div", [(1, 0), (2, 0), (3, 0), (5, 0), (2, 4), (2, 2), (3, 4), (4, 1), (3, 1000000)])
def test_threshold_rounds_bit_exact(engine_test, oracle, small_dangling, rounds, div):
    """Threshold rounds of the push (k_round_sweep; options "rounds" and "round_div"): 2^(rounds-1) x the threshold
    first, halved whenever a slot's frontier runs dry (div = 0) or is down to 1/div of the round's largest frontier.
    Every setting equals the twin running the same schedule bit for bit and ends with the exit condition of
    algo.h:1012; rounds that run dry never relax more edges than the plain schedule."""
    engine = engine_test  # the library build that has these schedules (conftest.py)
    g = small_dangling
    rmax, omega = _load(engine, g, epsilon=0.5)
    srcs = np.concatenate([pick_sources(g, 6, 71), pick_sources(g, 1, 72, want_dangling=True)])
    engine.set_option("rounds", 1)
    _, _, st1 = engine.push(srcs)
    engine.set_option("rounds", rounds)
    engine.set_option("round_div", div)
    oracle.twin_set_rounds(rounds)
    oracle.twin_set_round_div(div)
    try:
        rsv, res, st = engine.push(srcs)
        t1 = int(np.ceil(np.ldexp(rmax, 62)))
        for i, s in enumerate(srcs):
            t = oracle.twin_push(g, int(s), rmax)
            assert (res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all()
            assert st[i]["pops"] == t["pops"] and st[i]["relax"] == t["relax"] and st[i]["levels"] == t["levels"]
            assert int(rsv[i].sum()) + int(res[i].sum()) == oracle.FIX_ONE
            thr = (t1 * g.deg).astype(np.uint64)
            thr[g.deg == 0] = 1
            assert (res[i] < thr).all()
        if div == 0:
            assert sum(int(x["relax"]) for x in st) <= sum(int(x["relax"]) for x in st1)
        ppr, _, stq = engine.query_fix(srcs[:3], want_residue=False)
        for i in range(3):
            want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, seed=SEED)
            assert (ppr[i] == want).all() and stq[i]["n_walks"] == wst["n_walks"]
    finally:
        engine.reset_options()
        oracle.twin_set_rounds(1)
        oracle.twin_set_round_div(0)


@pytest.mark.parametrize("layout", ["narrow", "wide", "wide_multipass", "tail"])
@pytest.mark.parametrize("k,dmin", [(0, 0), (1, 0), (2, 0), (3, 0), (1, 300), (2, 3000)])
def test_bounded_deferral_bit_exact(engine_test, oracle, small_dangling, k, dmin, layout):
    """Bounded deferral of the push (option "defer", Dev::defer_k): a node that crosses with less than 2^k x its
    threshold waits one level.  Every k equals the twin running the same schedule bit for bit -- in the bitmap form of
    the bucketed levels, the list form of k_push_tail and across the hand-over between them -- ends with the exit
    condition of algo.h:1012, conserves mass exactly, and k = 1 relaxes fewer edges and leaves less residue than plain
    levels (k = 0, the default: the extra levels cost more on the GPU than the edges save).  dmin: only
    levels that pop at least that many nodes defer (option "defer_min")."""
    engine = engine_test  # the library build that has these schedules (conftest.py)
    g = small_dangling
    if layout in ("wide", "wide_multipass"):
        engine.set_option("force_wide", 1)
    if layout == "wide_multipass":
        engine.set_option("pass_bins", 1)
    if layout == "tail":
        engine.set_option("tail", 100000000)
        engine.set_option("tail_always", 1)
    else:
        engine.set_option("tail", 64)  # most levels bucketed, the small ones by k_push_tail: the hand-over is exercised
    rmax, omega = _load(engine, g, epsilon=0.5)
    srcs = np.concatenate([pick_sources(g, 6, 81), pick_sources(g, 1, 82, want_dangling=True)])
    try:
        engine.set_option("defer", 0)
        _, _, st0 = engine.push(srcs)
        engine.set_option("defer", k)
        engine.set_option("defer_min", dmin)
        oracle.twin_set_defer(k)
        oracle.twin_set_defer_min(dmin)
        rsv, res, st = engine.push(srcs)
        t1 = int(np.ceil(np.ldexp(rmax, 62)))
        for i, s in enumerate(srcs):
            t = oracle.twin_push(g, int(s), rmax)
            assert (res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all()
            assert st[i]["pops"] == t["pops"] and st[i]["relax"] == t["relax"] and st[i]["levels"] == t["levels"]
            assert int(rsv[i].sum()) + int(res[i].sum()) == oracle.FIX_ONE
            thr = (t1 * g.deg).astype(np.uint64)
            thr[g.deg == 0] = 1
            assert (res[i] < thr).all()
        if k == 1 and dmin == 0:
            assert sum(int(x["relax"]) for x in st) < sum(int(x["relax"]) for x in st0)
            assert sum(int(x["rsum_fix"]) for x in st) < sum(int(x["rsum_fix"]) for x in st0)
        ppr, _, stq = engine.query_fix(srcs[:3], want_residue=False)
        for i in range(3):
            want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, seed=SEED)
            assert (ppr[i] == want).all() and stq[i]["n_walks"] == wst["n_walks"]
    finally:
        engine.reset_options()
        oracle.twin_set_defer(0)
        oracle.twin_set_defer_min(0)


def test_two_lane_pipeline_same_bits(engine, oracle, small, monkeypatch):
    """Opt-in second lane (push of batch k+1 overlapping walks of batch k) gives the same bits."""
    g = small
    _load(engine, g, epsilon=0.5)
    srcs = pick_sources(g, 10, 61)
    engine.set_batch(3)
    a, ra, _ = engine.query_fix(srcs)
    engine.set_option("pipeline", 1)
    b, rb, st = engine.query_fix(srcs)
    engine.set_option("pipeline", 0)
    engine.set_batch(0)
    assert (a == b).all() and (ra == rb).all()
    assert all(s["ppr_sum_fix"] == oracle.FIX_ONE for s in st)


@pytest.mark.parametrize("gname", ["tiny_dangling", "small", "small_dangling"])
def test_power_iteration_bit_exact_vs_twin(engine, oracle, request, gname):
    """Exact SSPPR (gen_exact_topk's fwd_power_iteration, query.h:1192-1238) on the GPU: the level-capped push
    equals the twin bit for bit, agrees with the f64 restatement to 1e-12, and its top-k is the sorted head."""
    g = request.getfixturevalue(gname)
    _load(engine, g, epsilon=0.5)
    srcs = np.concatenate([pick_sources(g, 5, 81), pick_sources(g, 2, 82, want_dangling=True)])
    k = 50
    for iters in (100, 7):
        ppr, fix, ids, sc = engine.power_iteration(srcs, max_iter=iters, k=k, want_fix=True)
        for i, s in enumerate(srcs):
            want, st = oracle.twin_power_iteration(g, int(s), max_iter=iters)
            assert (fix[i] == want).all()
            assert (ppr[i] == oracle.fix_to_double(want)).all()
            nz = np.flatnonzero(want)
            order = nz[np.lexsort((nz, -(want[nz].astype(np.int64))))] if nz.size else nz
            head = order[:k]
            assert (ids[i][:head.size] == head).all()
            assert (sc[i][:head.size] == oracle.fix_to_double(want[head])).all()
            assert (ids[i][head.size:] == 0).all() and (sc[i][head.size:] == 0).all()
        if iters == 100:
            for i, s in enumerate(srcs[:3]):
                assert np.abs(ppr[i] - oracle.power_iteration(g, int(s), iters=100)).max() < 1e-12
    # the FORA answer honours its guarantee against this exact vector: |est - pi| <= eps * pi for pi >= 1/n
    est, _ = engine.query(srcs[:4])
    exact, _, _, _ = engine.power_iteration(srcs[:4], max_iter=100)
    for i in range(4):
        big = exact[i] >= 1.0 / g.n
        assert (np.abs(est[i][big] - exact[i][big]) <= 0.5 * exact[i][big]).all()
    import fora_amd
    with pytest.raises(fora_amd.ForaError):
        engine.power_iteration(srcs[:1], max_iter=0)


@pytest.mark.parametrize("with_idx", [False, True])
@pytest.mark.parametrize("k", [20, 500])
def test_topk_with_bounds_bit_exact_vs_twin(engine, oracle, small_dangling, k, with_idx):
    """get_topk without --opt (fora_query_topk_with_bound, query.h:909-969): rounds, ids and scores equal the
    twin's; k = 500 reaches the rounds where set_ppr_bounds / the bound-based stop rule are active."""
    g = small_dangling
    eps = 0.5
    rmax, omega = _load(engine, g, epsilon=eps)
    index = None
    if with_idx:
        engine.build_index()        # the non --opt index (build.h:328 else branch)
        index = engine.get_index()
    srcs = np.concatenate([pick_sources(g, 5, 95), pick_sources(g, 1, 96, want_dangling=True)])
    ids, sc, rounds = engine.topk_bound(srcs, k, epsilon=eps, with_idx=with_idx)
    for i, s in enumerate(srcs):
        wid, wsc, wr, _, _, _ = oracle.twin_topk_bound_query(g, int(s), k, eps, seed=SEED, index=index)
        assert rounds[i] == wr
        assert (ids[i] == wid).all()
        assert (sc[i] == wsc).all()
        if g.deg[s] > 0:
            exact = oracle.power_iteration(g, int(s))
            truth = set(np.argsort(-exact)[:k].tolist())
            assert len(truth & set(ids[i].tolist())) >= int(0.8 * k)
            rid, rsc, rr, _ = oracle.topk_bound_query(g, int(s), k, eps, seed=SEED, index=index)
            assert len(set(rid.tolist()) & set(ids[i].tolist())) >= int(0.8 * k) and abs(rr - wr) <= 1
        else:
            assert ids[i][0] == s and sc[i][0] == 1.0 and (sc[i][1:] == 0).all() and rounds[i] == 1
    # other decay exponent (moves `threshold`, query.h:913) and batching in slots of 2
    engine.set_batch(2)
    ids2, sc2, r2 = engine.topk_bound(srcs[:3], k, epsilon=eps, ppr_decay_alpha=0.5, with_idx=with_idx)
    engine.set_batch(0)
    for i, s in enumerate(srcs[:3]):
        wid, wsc, wr, _, _, _ = oracle.twin_topk_bound_query(g, int(s), k, eps, ppr_decay_alpha=0.5, seed=SEED, index=index)
        assert r2[i] == wr and (ids2[i] == wid).all() and (sc2[i] == wsc).all()
    engine.clear_index()


@pytest.mark.parametrize("with_idx", [False, True])
def test_balanced_bit_exact_vs_twin(engine, oracle, small_dangling, with_idx):
    """--balanced (fora_query_basic query.h:848-884 with the push charged by its work counters): rmax schedule, push
    state and refined ppr equal the twin's; the answer still honours the epsilon guarantee."""
    g = small_dangling
    rmax, omega = _load(engine, g, epsilon=0.5)
    index = None
    if with_idx:
        engine.build_index()
        index = engine.get_index()
    srcs = np.concatenate([pick_sources(g, 6, 97), pick_sources(g, 1, 98, want_dangling=True)])
    engine.set_balanced(True)
    try:
        ppr, res, st = engine.query_fix(srcs, with_idx=with_idx)
        engine.set_batch(3)
        ppr3, res3, st3 = engine.query_fix(srcs, with_idx=with_idx)
        engine.set_batch(0)
    finally:
        engine.set_balanced(False)
    assert (ppr3 == ppr).all() and (res3 == res).all()
    for i, s in enumerate(srcs):
        want, wres, wst = oracle.twin_query_balanced(g, int(s), rmax, omega, seed=SEED, index=index)
        assert (res[i] == wres).all() and (ppr[i] == want).all()
        assert st[i]["pops"] == wst["pops"] and st[i]["relax"] == wst["relax"] and st[i]["n_walks"] == wst["n_walks"]
        assert st[i]["ppr_sum_fix"] == oracle.FIX_ONE
        if g.deg[s] > 0:
            assert st[i]["push_rounds"] == wst["rounds"] >= 2 and st[i]["rmax_used"] == wst["rmax"]
            exact = oracle.power_iteration(g, int(s))
            big = exact >= 1.0 / g.n
            est = oracle.fix_to_double(ppr[i])
            assert (np.abs(est - exact)[big] / exact[big]).max() <= 0.5
        else:
            assert st[i]["dangling_source"] == 1 and st[i]["rmax_used"] == rmax
    # throughput flavour: first round at config.rmax
    engine.set_balanced(True, start_scale=1.0)
    p1, r1, s1 = engine.query_fix(srcs[:3], with_idx=with_idx)
    engine.set_balanced(False)
    for i, s in enumerate(srcs[:3]):
        want, wres, wst = oracle.twin_query_balanced(g, int(s), rmax, omega, seed=SEED, index=index, start_scale=1.0)
        assert (r1[i] == wres).all() and (p1[i] == want).all() and s1[i]["push_rounds"] == wst["rounds"]
    # plain mode reports one round at config.rmax
    _, _, st = engine.query_fix(srcs[:2])
    assert all(s["push_rounds"] == 1 and s["rmax_used"] == rmax for s in st)
    engine.clear_index()


def test_topk_select_compacted_form_same_lists(engine, oracle, small_dangling, monkeypatch):
    """Large graphs select the top k from an id-ordered compaction of each slot's non-zero entries; forced on a
    small graph it must give the very same lists (ties included) as the dense scan, in all three callers."""
    g = small_dangling
    _load(engine, g, epsilon=0.5, opt=True)
    srcs = np.concatenate([pick_sources(g, 5, 99), pick_sources(g, 1, 100, want_dangling=True)])
    out = {}
    for mode in ("0", "1"):
        engine.set_option("select_compact", int(mode))
        out[mode] = (engine.topk(srcs, 300, epsilon=0.5), engine.topk_bound(srcs, 300, epsilon=0.5),
                     engine.power_iteration(srcs, max_iter=3, k=700, want_ppr=False)[2:])
    engine.reset_options()
    for a, b in zip(out["0"], out["1"]):
        for x, y in zip(a, b):
            assert (np.asarray(x) == np.asarray(y)).all()
    # 3 iterations from a source reach few nodes: lists are zero-padded and tie-heavy (equal shares of one push)
    ids, sc = out["1"][2]
    assert (sc[:, -1] == 0).any() or (np.diff(sc, axis=1) == 0).any()


def test_schedule_experiments_are_not_in_the_product_library(engine, engine_test):
    """Threshold rounds and bounded deferral lost to the plain schedule and are compiled out of libfora_hip.so (the hot
    kernels do not carry their registers); the options exist in libfora_hip_test.so only."""
    from fora_amd.capi import ForaError
    assert engine.get_option("test_paths") == 0 and engine_test.get_option("test_paths") == 1
    for name, v in (("rounds", 2), ("round_div", 2), ("defer", 1), ("defer_min", 5)):
        with pytest.raises(ForaError) as e:
            engine.set_option(name, v)
        assert e.value.code == -1 and "not compiled" in str(e.value)
    for name, v in (("rounds", 1), ("round_div", 4), ("defer", 0), ("defer_min", 0)):
        engine.set_option(name, v)  # the defaults are accepted
