"""CPU tests of the oracle itself: fixtures of the reference, known-answer vectors,
an independent pure-Python restatement, invariants, and the epsilon guarantee."""
import os

import numpy as np
import pytest

import py_ref
from conftest import pick_sources

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- fixtures the reference ships (data/webstanford/attribute.txt, ssquery.txt)
def test_reference_attribute_fixture(oracle):
    n, m = oracle.read_attribute(os.path.join(GOLDEN, "webstanford_attribute.txt"))
    assert (n, m) == (281904, 2312497)


def test_reference_ssquery_fixture(oracle):
    q = oracle.read_queries(os.path.join(GOLDEN, "webstanford_ssquery.txt"))
    assert q.size == 1000
    assert q[0] == 103783 and q[1] == 91270 and q[2] == 135417
    assert q.min() == 361 and q.max() == 281501
    assert (q < 281904).all()


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert oracle.philox([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert py_ref.philox4x32_10([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def test_fora_setting_webstanford(oracle):
    # values the reference's own fora_setting printed for (n, m) of webstanford at eps=0.5
    # during the survey -- from a stub build, so a weak pin only
    rmax, omega = oracle.fora_setting(281904, 2312497, 0.5)
    assert repr(rmax) == "9.825025486429559e-08"
    assert repr(omega) == "37331049.94222769"
    assert (rmax, omega) == py_ref.fora_setting(281904, 2312497, 0.5)
    assert oracle.fora_setting(281904, 2312497, 0.5, opt=True) == py_ref.fora_setting(281904, 2312497, 0.5, opt=True)


# ---- loader semantics: graph.h:151-161
def test_loader_semantics(oracle, tmp_path):
    folder = tmp_path / "toy"
    folder.mkdir()
    (folder / "attribute.txt").write_text("n=5\nm=7\n")
    # self loop 2->2 dropped, duplicate 0->1 kept, file order kept per row
    (folder / "graph.txt").write_text("0 3\n0 1\n2 2\n0 1\n3 4\n1 0\n3 0\n")
    (folder / "ssquery.txt").write_text("0\n3\n4\n")
    g = oracle.Graph.from_folder(str(folder))
    assert (g.n, g.m) == (5, 7)  # m comes from attribute.txt, not recounted
    assert g.row_ptr.tolist() == [0, 3, 4, 4, 6, 6]
    assert g.col.tolist() == [3, 1, 1, 0, 4, 0]
    assert oracle.read_queries(str(folder / "ssquery.txt")).tolist() == [0, 3, 4]
    with pytest.raises(ValueError):
        oracle.Graph.from_edges(3, 1, [0], [3])


def _adj(g):
    return [g.col[g.row_ptr[v]:g.row_ptr[v + 1]].tolist() for v in range(g.n)]


# ---- C oracle vs the independent pure-Python restatement, bit for bit
@pytest.mark.parametrize("gname", ["tiny", "tiny_dangling"])
def test_push_fifo_matches_python_restatement(oracle, request, gname):
    g = request.getfixturevalue(gname)
    adj = _adj(g)
    rmax, omega = oracle.fora_setting(g.n, g.m, 0.5)
    srcs = list(pick_sources(g, 4, 1)) + list(pick_sources(g, 1, 2, want_dangling=True))
    for s in srcs:
        p = oracle.push_fifo(g, int(s), rmax)
        rsv, res, rsum = py_ref.push_fifo(adj, int(s), rmax)
        assert p["rsum"] == rsum
        assert p["reserve_occur"].tolist() == list(rsv.keys())
        assert p["residue_occur"].tolist() == list(res.keys())
        assert [p["reserve"][v] for v in rsv] == list(rsv.values())
        assert [p["residue"][v] for v in res] == list(res.values())
        for opt in (False, True):
            N, cnt = oracle.walk_counts(p, omega, opt=opt)
            if rsum == 0:
                assert N == 0
                continue
            N2, cnt2 = py_ref.walk_counts(res, rsum, omega, opt=opt)
            assert N == N2 and cnt.tolist() == cnt2


def test_walk_matches_python_restatement(oracle, tiny_dangling):
    g = tiny_dangling
    adj = _adj(g)
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(300):
        v = int(rng.integers(0, g.n))
        j = int(rng.integers(0, 1 << 40))
        nzh = bool(rng.integers(0, 2))
        rnd = int(rng.integers(0, 4))
        assert oracle.walk(g, 0x464F5241, 77, rnd, v, j, no_zero_hop=nzh) == \
            py_ref.walk(adj, 0x464F5241, 77, rnd, v, j, no_zero_hop=nzh)


def test_long_walks_leave_the_512_step_period(oracle, tiny):
    """The step field of the Philox counter has 8 bits (one call per two steps): without the (step >> 9) term in the
    last counter word a walk that survives 512 steps would replay its draws for ever.  alpha = 0.004 makes such walks
    common (0.996^512 = 13 %); both restatements agree on them and none runs away."""
    g = tiny
    adj = _adj(g)
    rng = np.random.Generator(np.random.PCG64(6))
    long_ones = 0
    for _ in range(60):
        v = int(rng.choice(np.flatnonzero(g.deg > 0)))
        j = int(rng.integers(0, 1 << 40))
        steps = oracle.walk_steps(g, 0x464F5241, 3, 0, v, j, alpha=0.004)
        long_ones += steps > 512
        assert steps < 20000
        assert oracle.walk(g, 0x464F5241, 3, 0, v, j, alpha=0.004) == py_ref.walk(adj, 0x464F5241, 3, 0, v, j, alpha=0.004)
    assert long_ones >= 2


# ---- invariants of both families (last row)
@pytest.mark.parametrize("gname", ["small", "small_dangling"])
def test_push_invariants(oracle, request, gname):
    g = request.getfixturevalue(gname)
    rmax, _ = oracle.fora_setting(g.n, g.m, 0.5)
    deg = g.deg
    for s in pick_sources(g, 3, 3):
        p = oracle.push_fifo(g, int(s), rmax)
        res = np.where(p["residue"] < 0, 0, p["residue"])
        rsv = np.where(p["reserve"] < 0, 0, p["reserve"])
        assert abs(res.sum() + rsv.sum() - 1) < 1e-12
        assert abs(res.sum() - p["rsum"]) < 1e-12
        assert (res[deg > 0] / deg[deg > 0] < rmax).all()     # exit condition algo.h:1012
        assert (res[deg == 0] == 0).all()                     # dangling nodes always pushed
        t = oracle.twin_push(g, int(s), rmax)
        assert int(t["residue"].sum()) + int(t["reserve"].sum()) == oracle.FIX_ONE  # exact mass
        assert int(t["residue"].sum()) == t["rsum_fix"]
        t1 = oracle.lib().orc_twin_rmax_fix(__import__("ctypes").c_double(rmax))
        nz = deg > 0
        assert (t["residue"][nz].astype(object) < t1 * deg[nz].astype(object)).all()
        assert (t["residue"][~nz] == 0).all()
        assert t["pops"] >= 1 and t["levels"] >= 1


def test_dangling_source(oracle, small_dangling):
    g = small_dangling
    rmax, omega = oracle.fora_setting(g.n, g.m, 0.5)
    s = int(pick_sources(g, 1, 4, want_dangling=True)[0])
    ppr, st = oracle.query(g, s, rmax, omega)
    assert st["rsum"] == 0 and st["n_walks"] == 0 and ppr[s] == 1 and ppr.sum() == 1  # algo.h:961-965
    pf, rf, st2 = oracle.twin_query(g, s, rmax, omega)
    assert pf[s] == oracle.FIX_ONE and int(pf.sum()) == oracle.FIX_ONE and st2["n_walks"] == 0


# ---- accuracy: both families meet the FORA guarantee against exact PPR (query.h:1192-1224)
@pytest.mark.parametrize("opt", [False, True])
@pytest.mark.parametrize("with_idx", [False, True])
def test_epsilon_guarantee_and_family_gap(oracle, small, opt, with_idx):
    g = small
    eps = 0.5
    rmax, omega = oracle.fora_setting(g.n, g.m, eps, opt=opt)
    index = oracle.build_index(g, 11, rmax, omega, opt=opt) if with_idx else None
    for s in pick_sources(g, 2, 5):
        s = int(s)
        exact = oracle.power_iteration(g, s)
        a, st = oracle.query(g, s, rmax, omega, opt=opt, seed=11, index=index)
        bf, _, st2 = oracle.twin_query(g, s, rmax, omega, opt=opt, seed=11, index=index)
        b = oracle.fix_to_double(bf)
        assert abs(a.sum() - 1) < 1e-9 and int(bf.sum()) == oracle.FIX_ONE
        big = exact >= 1.0 / g.n
        for est in (a, b):
            assert (np.abs(est - exact)[big] / exact[big]).max() <= eps
        # the two schedules end in different (reserve, residue) pairs; the refined vectors
        # agree to Monte-Carlo noise: stated L-inf gap 1e-3 at this size
        assert np.abs(a - b).max() < 1e-3
        if with_idx:
            assert st["n_idx_hit"] == st["n_walks"] and st2["n_idx_hit"] == st2["n_walks"]  # 100 % hit


def test_index_sizes_and_contents(oracle, tiny):
    g = tiny
    rmax, omega = oracle.fora_setting(g.n, g.m, 0.5)
    total, off, cnt = oracle.index_sizes(g, rmax, omega)
    assert total == cnt.sum() and off[0] == 0 and (np.diff(off.astype(np.int64)) == cnt[:-1].astype(np.int64)).all()
    assert (cnt == np.ceil(g.deg * rmax * omega)).all()       # build.h:331
    rw, off2, cnt2 = oracle.build_index(g, 3, rmax, omega)
    assert rw.size == total and rw.min() >= 0 and rw.max() < g.n
    v = int(np.argmax(cnt))
    assert rw[off[v]] == oracle.walk(g, 3, oracle.STREAM_INDEX, 0, v, 0)


def test_topk_families_agree(oracle, small):
    g = small
    k = 20
    for with_idx in (False, True):
        index = None
        if with_idx:
            rmax, omega = oracle.fora_setting(g.n, g.m, 0.5, opt=True)
            index = oracle.build_index(g, 9, rmax, omega, opt=True)
        for s in pick_sources(g, 2, 6):
            s = int(s)
            exact = oracle.power_iteration(g, s)
            truth = set(np.argsort(-exact)[:k].tolist())
            ids, sc, rounds, _ = oracle.topk_query(g, s, k, 0.5, seed=9, index=index)
            ids2, sc2, rounds2, _ = oracle.twin_topk_query(g, s, k, 0.5, seed=9, index=index)
            assert 1 <= rounds <= 12 and 1 <= rounds2 <= 12
            assert (np.diff(sc) <= 0).all() and (np.diff(sc2) <= 0).all()
            assert len(truth & set(ids.tolist())) >= k - 3
            assert len(truth & set(ids2.tolist())) >= k - 3
            assert np.abs(sc[0] - sc2[0]) / sc[0] < 0.05


@pytest.mark.parametrize("gname", ["tiny", "tiny_dangling"])
def test_power_iteration_families_agree(oracle, request, gname):
    """fwd_power_iteration (query.h:1192-1224): the f64 restatement, the fixed-point twin (level-capped push with
    the smallest threshold) and a dense numpy iteration give the same vector; dangling sources iterate too."""
    g = request.getfixturevalue(gname)
    srcs = list(pick_sources(g, 3, 71)) + list(pick_sources(g, 1, 72, want_dangling=True))
    for s in srcs:
        s = int(s)
        want = oracle.power_iteration(g, s, iters=100)
        fix, st = oracle.twin_power_iteration(g, s, max_iter=100)
        got = oracle.fix_to_double(fix)
        assert st["levels"] <= 100
        assert int(fix.sum()) + st["rsum_fix"] == oracle.FIX_ONE           # mass is conserved exactly
        assert np.abs(got - want).max() < 1e-12
        assert abs(got.sum() - (1 - 0.8 ** 100)) < 1e-9                    # what 100 iterations reserve
        # dense numpy restatement
        r = np.zeros(g.n); r[s] = 1.0
        p = np.zeros(g.n)
        deg = g.deg.astype(np.float64)
        rows = np.repeat(np.arange(g.n), g.deg)
        for _ in range(100):
            p += 0.2 * r
            push = np.where(g.deg > 0, 0.8 * r / np.maximum(deg, 1), 0.0)
            nxt = np.bincount(g.col, weights=push[rows], minlength=g.n)
            nxt[s] += 0.8 * r[g.deg == 0].sum()
            r = nxt
        assert np.abs(p - want).max() < 1e-12


SEED = 0x464F5241
_GD = {}


def oracle_graph_dangling(oracle):
    if "g" not in _GD:
        from fora_amd import synth
        n, m, seed = synth.PRESETS["tiny"]
        src, dst = synth.rmat_graph(n, m, seed, "rmat")
        _GD["g"] = oracle.Graph.from_edges(n, m, src, dst)
    return _GD["g"]


def test_topk_with_bounds_families(oracle, small):
    """get_topk without --opt (query.h:909-969): the reference-order restatement and the twin agree on the round
    schedule and on (nearly) all of the list, the list is accurate, and the bounds bracket the exact PPR."""
    g = small
    eps = 0.5
    srcs = pick_sources(g, 2, 91)
    for k in (20, 500):
        for s in srcs:
            s = int(s)
            ids, sc, rounds, _ = oracle.topk_bound_query(g, s, k, eps, seed=SEED)
            ti, ts, tr, tp, up, lo = oracle.twin_topk_bound_query(g, s, k, eps, seed=SEED, want_ppr=True, want_bounds=True)
            assert abs(rounds - tr) <= 1
            assert (np.diff(sc) <= 0).all() and (np.diff(ts) <= 0).all()
            assert len(set(ids.tolist()) & set(ti.tolist())) >= int(0.9 * k)
            exact = oracle.power_iteration(g, s)
            truth = set(np.argsort(-exact)[:k].tolist())
            assert len(truth & set(ti.tolist())) >= int(0.85 * k)
            assert len(truth & set(ids.tolist())) >= int(0.85 * k)
            # bounds hold with probability 1 - pfail per node with pi >= 1/n (the relative bound of
            # algo.h:1183-1185 assumes pi >= min_ppr)
            big = exact >= 1.0 / g.n
            assert (up[big] >= exact[big]).all()
            assert (lo[big] <= exact[big]).all()
            assert int(tp.sum()) == oracle.FIX_ONE
    # k = 500 runs into the rounds below `threshold`, where the bounds are maintained
    assert (lo > 0).any() and (up < 1).any()
    # dangling source: one round, ppr = e_s (query.h:951-955)
    d = int(pick_sources(oracle_graph_dangling(oracle), 1, 92, want_dangling=True)[0])
    gd = oracle_graph_dangling(oracle)
    ids, sc, rounds, _ = oracle.topk_bound_query(gd, d, 10, eps, seed=SEED)
    ti, ts, tr, _, _, _ = oracle.twin_topk_bound_query(gd, d, 10, eps, seed=SEED)
    assert rounds == tr == 1 and ids[0] == ti[0] == d and sc[0] == ts[0] == 1.0 and (sc[1:] == 0).all()


def test_calculate_lambda_operand_order(oracle):
    """algo.h:1169-1174 against the same expression written in numpy."""
    rsum, pfail, ub, tot = 0.23, 1e-11, 0.4, 123456
    L = np.log(2 / pfail)
    want = 1.0 / 3 * L * rsum / tot + np.sqrt(4.0 / 9.0 * L * L * rsum * rsum + 8 * tot * L * rsum * ub) / 2.0 / tot
    assert oracle.calculate_lambda(rsum, pfail, ub, tot) == want


def test_twin_balanced_mode(oracle, small):
    """--balanced in the twin: conserves mass, keeps the guarantee, ends at a smaller rmax with fewer walks than
    the plain run when walks are the expensive side, and degenerates to 8*rmax when they are free."""
    g = small
    rmax, omega = oracle.fora_setting(g.n, g.m, 0.5)
    for s in pick_sources(g, 2, 93):
        s = int(s)
        ppr0, _, st0 = oracle.twin_query(g, s, rmax, omega, seed=SEED)
        ppr, res, st = oracle.twin_query_balanced(g, s, rmax, omega, seed=SEED)
        assert int(ppr.sum()) == oracle.FIX_ONE
        assert st["rounds"] >= 2 and st["rmax"] == rmax * 8 / 2 ** (st["rounds"] - 1)
        exact = oracle.power_iteration(g, s)
        big = exact >= 1.0 / g.n
        est = oracle.fix_to_double(ppr)
        assert (np.abs(est - exact)[big] / exact[big]).max() <= 0.5
        if st["rmax"] < rmax:
            assert st["n_walks"] < st0["n_walks"] and st["relax"] > st0["relax"]
        _, _, free = oracle.twin_query_balanced(g, s, rmax, omega, seed=SEED, t_walk=1e-30)
        assert free["rounds"] == 1 and free["rmax"] == rmax * 8
        _, _, one = oracle.twin_query_balanced(g, s, rmax, omega, seed=SEED, start_scale=1.0)
        assert one["rmax"] == rmax / 2 ** (one["rounds"] - 1) and one["rounds"] < st["rounds"]


@pytest.mark.parametrize("rounds,div", [(2, 0), (3, 0), (2, 4), (3, 2)])
def test_twin_threshold_rounds_keep_the_push_invariants(oracle, small_dangling, rounds, div):
    """Threshold rounds of the twin (the engine's options "rounds" / "round_div"): whatever the schedule, the push ends
    with every residue under its threshold (algo.h:1012), mass is conserved exactly, and the plain schedule is what the
    other tests pin -- it is restored afterwards."""
    g = small_dangling
    rmax, omega = oracle.fora_setting(g.n, g.m, 0.5)
    t1 = int(np.ceil(np.ldexp(rmax, 62)))
    thr = (t1 * g.deg).astype(np.uint64)
    thr[g.deg == 0] = 1
    srcs = pick_sources(g, 4, 91)
    plain = [oracle.twin_push(g, int(s), rmax) for s in srcs]
    oracle.twin_set_rounds(rounds)
    oracle.twin_set_round_div(div)
    try:
        for s, p in zip(srcs, plain):
            t = oracle.twin_push(g, int(s), rmax)
            assert (t["residue"] < thr).all()
            assert int(t["reserve"].sum()) + int(t["residue"].sum()) == oracle.FIX_ONE
            assert t["rsum_fix"] == int(t["residue"].sum())
            if div == 0:
                assert t["relax"] <= p["relax"]  # rounds that run dry never relax more edges
    finally:
        oracle.twin_set_rounds(1)
        oracle.twin_set_round_div(0)
    again = oracle.twin_push(g, int(srcs[0]), rmax)
    assert (again["residue"] == plain[0]["residue"]).all() and again["relax"] == plain[0]["relax"]


def test_query_many_matches_single_queries(oracle, tiny):
    """orc_query_many (bench.py's all-core CPU baseline): the query loop on several pthreads with private buffers does the
    same work as one orc_query per source."""
    from fora_amd import synth
    g = tiny
    rmax, omega = oracle.fora_setting(g.n, g.m, 0.5)
    srcs = synth.query_set(g.n, 24, 7)
    walks = sum(oracle.query(g, int(s), rmax, omega, seed=3)[1]["n_walks"] for s in srcs)
    for threads in (1, 3):
        done, dt, w = oracle.query_many(g, srcs, rmax, omega, threads, 60.0, seed=3)
        assert done == len(srcs) and w == walks and dt > 0
    done, _, _ = oracle.query_many(g, srcs, rmax, omega, 2, 0.0, seed=3)  # budget spent at once: one query per thread
    assert done == 2


def test_topk_push_counts_follow_the_rounds(oracle, small):
    """orc_topk_push_counts (bench.py's roofline of the top-k configuration): the pushes of the first r rounds of the --opt
    driver without the walks between them.  The push state does not depend on the walks, so the counts of r rounds are those
    of r - 1 rounds plus one more push; a dangling source pushes nothing; the full run's round count is a valid argument."""
    g = small
    s = int(np.flatnonzero(g.deg > 0)[11])
    _, _, rounds, _ = oracle.topk_query(g, s, 50, 0.5, seed=7)
    assert rounds >= 1
    prev = (0, 0)
    for r in range(1, rounds + 2):
        cur = oracle.topk_push_counts(g, s, 50, 0.5, r)
        assert cur[0] >= prev[0] and cur[1] >= prev[1]
        prev = cur
    assert prev[0] > 0 and prev[1] > 0
    one = oracle.topk_push_counts(g, s, 50, 0.5, 1)
    fifo = oracle.push_fifo(g, s, oracle.fora_topk_setting(g.m, 0.5, 1.0 / 50 / 10, 1.0 / g.n / g.n)[0]) if hasattr(oracle, "fora_topk_setting") else None
    if fifo is not None:  # round 1 is a plain FIFO push from the source at the round's rmax: same pops and relaxations
        assert one == (fifo["pops"], fifo["relax"])
