"""BASELINE.json configs 3-5 at full size, through the C ABI, on graphs built by the HIP R-MAT generator
(tools/rmat_gen.hip):

  config 3  LiveJournal-sized (n = 4 847 571, m = 68 993 773), eps = 0.5, --with_idx, 1 x MI355X
  config 4  Twitter-2010-sized (n = 41 652 230, m = 1 468 365 182), eps = 0.5, --with_idx  (per-GPU shard of the 8-GPU run)
  config 5  Twitter-2010-sized, topk k = 500 --opt --with_idx

What is checked at these sizes: index sizes against the oracle (build.h:325-334), bit-exact push / query of one source
against the CPU twin (the only full-size oracle run that finishes in seconds to a minute), and the size-independent
properties for a whole batch: exact mass conservation, 100 % index hit, the push's exit condition over
ALL nodes (algo.h:1012), sorted top-k lists and the round bound of query.h:972-1045.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x464F5241


def _exit_condition_holds(residue, deg, rmax):
    """residue[v] < ceil(rmax * 2^62) * outdeg(v) for every node; dangling nodes hold no residue (algo.h:1012)."""
    t1 = np.uint64(int(np.ceil(np.ldexp(rmax, 62))))
    thr = t1 * deg.astype(np.uint64)  # < 2^63 for these graphs (t1 ~ 1e9..1e10, degrees < 1e7)
    thr[deg == 0] = 1
    return bool((residue < thr).all())


def test_generator_contract(engine):
    """The HIP generator: exactly m distinct non-loop edges, rows sorted, every node an out-edge in mode "none",
    plain R-MAT leaves dangling nodes, same arguments -> same graph."""
    from fora_amd import synth
    n, m = 5000, 40000
    rp, col = synth.rmat_csr_gpu(n, m, 77, "none")
    assert rp[0] == 0 and rp[-1] == m and col.shape == (m,)
    deg = np.diff(rp)
    assert deg.min() >= 1
    src = np.repeat(np.arange(n), deg)
    assert (src != col).all() and col.min() >= 0 and col.max() < n
    keys = src.astype(np.int64) * n + col
    assert (np.diff(keys) > 0).all()  # sorted by (source, target) and distinct
    rp2, col2 = synth.rmat_csr_gpu(n, m, 77, "none")
    assert (rp2 == rp).all() and (col2 == col).all()
    rp3, col3 = synth.rmat_csr_gpu(n, m, 78, "none")
    assert (col3 != col).any()
    rp4, col4 = synth.rmat_csr_gpu(n, m, 77, "rmat")
    assert rp4[-1] == m and (np.diff(rp4) == 0).mean() > 0.2
    # degree skew of R-MAT: the heaviest 1 % of the nodes own far more than 1 % of the in-edges
    indeg = np.bincount(col4, minlength=n)
    assert np.sort(indeg)[-n // 100:].sum() > 0.1 * m


@pytest.fixture(scope="module")
def medium(oracle, engine):
    from fora_amd import synth
    n, m, rp, col = synth.preset("medium")
    return oracle.Graph(n, m, rp, col)


def test_medium_wide_layout_bit_exact(engine, oracle, medium):
    """184 bins: the wide bucket layout (12-byte messages, indexed walk results chunk-binned) at a size where the
    twin still answers in seconds -- push, indexed query and top-k bit for bit."""
    g = medium
    engine.clear_index()
    engine.set_graph(g.n, g.m, g.row_ptr, g.col)
    engine.set_params(epsilon=0.5, seed=SEED)
    rmax, omega = engine.get_params()
    assert (rmax, omega) == oracle.fora_setting(g.n, g.m, 0.5)
    from fora_amd import synth
    srcs = synth.query_set(g.n, 3, 3)
    rsv, res, st = engine.push(srcs)
    for i, s in enumerate(srcs):
        t = oracle.twin_push(g, int(s), rmax)
        assert (res[i] == t["residue"]).all() and (rsv[i] == t["reserve"]).all()
        assert st[i]["pops"] == t["pops"] and st[i]["relax"] == t["relax"] and st[i]["levels"] == t["levels"]
        assert _exit_condition_holds(res[i], g.deg, rmax)
    # the bin kernel of this layout reads quad-padded copies of col / col_hub (Dev::col4, round 5); with option quads = 0 it
    # reads single edges as the multi-pass layouts still do: same bits
    engine.set_option("quads", 0)
    rsv1, res1, _ = engine.push(srcs)
    engine.set_option("quads", 1)
    assert (rsv1 == rsv).all() and (res1 == res).all()
    del rsv1, res1
    total, off, cnt = engine.index_sizes()
    t2, off2, cnt2 = oracle.index_sizes(g, rmax, omega)
    assert total == t2 and (off == off2).all() and (cnt == cnt2).all()
    engine.build_index()
    idx = engine.get_index()
    ppr, _, st = engine.query_fix(srcs[:2], with_idx=True, want_residue=False)
    for i in range(2):
        want, _, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, seed=SEED, index=idx)
        assert (ppr[i] == want).all()
        assert st[i]["n_idx_hit"] == st[i]["n_walks"] == wst["n_walks"]
    engine.clear_index()


def test_webstanford_size_rmat_dangling(engine, oracle):
    """The plain R-MAT variant of the headline graph: about 43 % of the nodes have no out-edge, so
    dangling targets send their mass back to the source (algo.h:993-999, the hot `residue[s] +=` path) in nearly
    every level, and dangling sources return at once (algo.h:961-965)."""
    from fora_amd import synth
    n, m, rp, col = synth.preset("webstanford", "rmat")
    g = oracle.Graph(n, m, rp, col)
    deg = g.deg
    assert 0.3 < (deg == 0).mean() < 0.6
    engine.clear_index()
    engine.set_graph(n, m, rp, col)
    engine.set_params(epsilon=0.5, seed=SEED)
    rmax, omega = engine.get_params()
    srcs = synth.query_set(n, 48, 17)
    assert (deg[srcs] == 0).any() and (deg[srcs] > 0).any()
    ppr, res, st = engine.query_fix(srcs[:12])
    for i in range(12):
        s = int(srcs[i])
        assert int(ppr[i].sum()) == 1 << 62 and st[i]["ppr_sum_fix"] == 1 << 62
        if deg[s] == 0:
            assert st[i]["dangling_source"] == 1 and ppr[i][s] == 1 << 62 and st[i]["n_walks"] == 0   # ppr = e_s
        else:
            assert _exit_condition_holds(res[i], deg, rmax)
            assert (res[i][deg == 0] == 0).all()                            # a dangling node never keeps residue
    for i in [int(j) for j in np.flatnonzero(deg[srcs[:12]] > 0)[:2]]:     # two sources bit for bit against the twin
        want, wres, wst = oracle.twin_query(g, int(srcs[i]), rmax, omega, seed=SEED)
        assert (res[i] == wres).all() and (ppr[i] == want).all()
        assert st[i]["pops"] == wst["pops"] and st[i]["relax"] == wst["relax"] and st[i]["levels"] == wst["levels"]
        assert st[i]["n_walks"] == wst["n_walks"]
    _, st = engine.query(srcs, want_ppr=False)
    assert all(s["ppr_sum_fix"] == 1 << 62 for s in st)


def test_livejournal_with_idx(engine, oracle):
    """BASELINE config 3."""
    from fora_amd import synth
    t0 = time.time()
    n, m, rp, col = synth.preset("livejournal")
    t_gen = time.time() - t0
    assert t_gen < 60, f"graph generation took {t_gen:.0f} s"
    g = oracle.Graph(n, m, rp, col)
    deg = g.deg
    engine.clear_index()
    engine.set_graph(n, m, rp, col)
    engine.set_params(epsilon=0.5, seed=SEED)
    rmax, omega = engine.get_params()
    assert (rmax, omega) == oracle.fora_setting(n, m, 0.5)
    total, off, cnt = engine.index_sizes()
    t2, off2, cnt2 = oracle.index_sizes(g, rmax, omega)
    assert total == t2 and (off == off2).all() and (cnt == cnt2).all()   # build.h:325-334
    assert 1.9e8 < total < 2.4e8                                          # ~2.14e8 entries
    engine.build_index()
    srcs = synth.query_set(n, 96, 11)
    ppr, res, st = engine.query_fix(srcs[:8], with_idx=True)
    for i in range(8):
        assert int(ppr[i].sum()) == 1 << 62 and st[i]["ppr_sum_fix"] == 1 << 62
        assert int(res[i].sum()) == st[i]["rsum_fix"]
        assert _exit_condition_holds(res[i], deg, rmax)                   # over ALL nodes
        assert st[i]["n_idx_hit"] == st[i]["n_walks"] > 0                 # 100 % index hit
    # one source bit for bit against the twin, with the engine's own index
    idx = engine.get_index()
    want, wres, wst = oracle.twin_query(g, int(srcs[0]), rmax, omega, seed=SEED, index=idx)
    assert (res[0] == wres).all() and (ppr[0] == want).all()
    assert st[0]["pops"] == wst["pops"] and st[0]["relax"] == wst["relax"] and st[0]["n_walks"] == wst["n_walks"]
    # the same source against the REFERENCE-ORDER restatement and against exact PPR, at this size (round 5): the FIFO oracle's
    # query with the engine's index (f64, pop order of algo.h:980-1017: another valid push state, the same indexed walks for
    # the nodes both leave residue on) and the f64 power iteration of query.h:1192-1224 on the CPU.  Tolerances: L-inf between
    # the GPU's and the FIFO oracle's estimates <= 2e-4 (two push states + walk noise; measured ~1e-5), and both inside the
    # FORA guarantee |est - pi| <= eps * pi for pi >= 1/n (algo.h:455-463) -- the GPU's also against the GPU power iteration.
    est_gpu = oracle.fix_to_double(ppr[0])
    est_fifo, fst = oracle.query(g, int(srcs[0]), rmax, omega, seed=SEED, index=idx)
    assert np.abs(est_gpu - est_fifo).max() <= 2e-4
    assert abs(float(est_fifo.sum()) - 1.0) < 1e-9
    exact = oracle.power_iteration(g, int(srcs[0]))                       # CPU, f64, 100 iterations
    big = exact >= 1.0 / n
    assert big.sum() > 100
    rel_gpu = float((np.abs(est_gpu - exact)[big] / exact[big]).max())
    rel_fifo = float((np.abs(est_fifo - exact)[big] / exact[big]).max())
    assert rel_gpu <= 0.5 and rel_fifo <= 0.5, (rel_gpu, rel_fifo)        # eps = 0.5
    gexact, _, _, _ = engine.power_iteration(srcs[:1], max_iter=100)
    assert np.abs(gexact[0] - exact).max() <= 1e-12
    print(f"LJ-sized source {int(srcs[0])}: L-inf GPU vs FIFO oracle {np.abs(est_gpu - est_fifo).max():.2e}, worst rel. err on pi >= 1/n: GPU {rel_gpu:.3f}, FIFO {rel_fifo:.3f}")
    del est_fifo, exact, gexact
    # the whole batch of the config's flavour: stats only (results stay in HBM)
    _, st = engine.query(srcs, with_idx=True, want_ppr=False)
    assert len(st) == 96
    assert all(s["ppr_sum_fix"] == 1 << 62 for s in st)
    assert all(s["n_idx_hit"] == s["n_walks"] for s in st)
    assert st[0]["n_walks"] == wst["n_walks"]                             # batch position does not matter
    engine.clear_index()
    # ---- top-k at this size: k = 500 --opt --with_idx (query.h:972-1045), two sources bit for bit against the twin
    del idx, ppr, res, want, wres
    engine.set_params(epsilon=0.5, opt=True, seed=SEED)
    engine.build_index()
    idx = engine.get_index()
    ids, sc, rounds = engine.topk(srcs[:4], 500, epsilon=0.5, with_idx=True)
    assert (np.diff(sc, axis=1) <= 0).all() and (sc[:, 0] > 0).all()
    for i in range(2):
        wid, wsc, wr, _ = oracle.twin_topk_query(g, int(srcs[i]), 500, 0.5, seed=SEED, index=idx)
        assert rounds[i] == wr and (ids[i] == wid).all() and (sc[i] == wsc).all()
    engine.clear_index()


def test_twitter2010_with_idx_and_topk(engine, oracle):
    """BASELINE configs 4 and 5 (one GPU's shard)."""
    from fora_amd import synth
    t0 = time.time()
    n, m, rp, col = synth.preset("twitter2010")
    t_gen = time.time() - t0
    assert t_gen < 240, f"graph generation took {t_gen:.0f} s"
    assert rp[-1] == m == 1_468_365_182
    deg = np.diff(rp)
    engine.clear_index()
    engine.set_graph(n, m, rp, col)
    engine.set_params(epsilon=0.5, seed=SEED)
    rmax, omega = engine.get_params()
    assert (rmax, omega) == oracle.fora_setting(n, m, 0.5)
    srcs = synth.query_set(n, 24, 13)
    g = oracle.Graph(n, m, rp, col)
    # ---- config 4: indexed queries.  One source bit for bit against the twin -- push (residue, pops, relaxations, levels)
    # AND the indexed refinement (query.h:276-308) over the engine's own index: about a minute and a half of one core
    total, _, _ = engine.index_sizes()
    assert 2.8e9 < total < 3.4e9                                          # ~3.07e9 entries (11.4 GiB)
    engine.build_index()
    idx = engine.get_index()
    ppr, res, st = engine.query_fix(srcs[:1], with_idx=True)
    want, wres, wst = oracle.twin_query(g, int(srcs[0]), rmax, omega, seed=SEED, index=idx)
    assert (res[0] == wres).all() and (ppr[0] == want).all()
    assert st[0]["pops"] == wst["pops"] and st[0]["relax"] == wst["relax"] and st[0]["levels"] == wst["levels"]
    assert st[0]["n_walks"] == wst["n_walks"] == st[0]["n_idx_hit"] and int(ppr[0].sum()) == 1 << 62
    assert _exit_condition_holds(res[0], deg, rmax)
    del ppr, res, want, wres, idx
    _, st = engine.query(srcs, with_idx=True, want_ppr=False)
    assert all(s["ppr_sum_fix"] == 1 << 62 for s in st)                   # mass conserved exactly
    assert all(s["n_idx_hit"] == s["n_walks"] > 0 for s in st)            # 100 % index hit
    assert all(s["levels"] > 0 and s["pops"] > 0 for s in st if not s["dangling_source"])
    assert st[0]["n_walks"] == wst["n_walks"]                             # batch position does not matter
    # ---- config 5: topk k = 500 --opt --with_idx (the --opt index: one-hop walks, build.h:328-329)
    engine.clear_index()
    engine.set_params(epsilon=0.5, opt=True, seed=SEED)
    engine.build_index()
    ids, sc, rounds = engine.topk(srcs[:8], 500, epsilon=0.5, with_idx=True)
    assert ids.shape == (8, 500)
    assert (np.diff(sc, axis=1) <= 0).all() and (sc[:, 0] > 0).all()      # sorted, non-empty
    assert (rounds >= 1).all() and (rounds <= 8).all()                    # query.h:1001: delta from 1/(10k) down to 1/n in /4 steps
    for i in range(8):
        k_pos = int((sc[i] > 0).sum())
        assert len(set(ids[i, :k_pos].tolist())) == k_pos                 # no node twice
        assert (ids[i, :k_pos] >= 0).all() and (ids[i, :k_pos] < n).all()
    ids2, sc2, rounds2 = engine.topk(srcs[:8], 500, epsilon=0.5, with_idx=True)
    assert (ids2 == ids).all() and (sc2 == sc).all() and (rounds2 == rounds).all()   # reproducible
    # two sources bit for bit against the twin's top-k driver (ids, scores, rounds) with the engine's --opt index
    idx = engine.get_index()
    for i in range(2):
        wid, wsc, wr, _ = oracle.twin_topk_query(g, int(srcs[i]), 500, 0.5, seed=SEED, index=idx)
        assert rounds[i] == wr and (ids[i] == wid).all() and (sc[i] == wsc).all()
    del idx
    engine.clear_index()
