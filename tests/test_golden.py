"""Committed golden vectors (tests/golden/oracle_vectors.json, made by tests/golden/make_golden.py):
CPU test -- the oracle still reproduces them; GPU test -- the HIP path reproduces them without
consulting the oracle at run time."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cases():
    return json.load(open(GOLDEN))


def _graph(case):
    from fora_amd import synth
    n, m, seed = synth.PRESETS[case["graph"]]
    src, dst = synth.rmat_graph(n, m, seed, case["dangling"])
    return n, m, synth.csr_from_edges(n, src, dst)


def test_oracle_reproduces_golden(oracle):
    G = _cases()
    for case in G["cases"]:
        n, m, (row_ptr, col) = _graph(case)
        g = oracle.Graph(n, m, row_ptr, col)
        assert sha(g.row_ptr) + sha(g.col) == case["csr_sha256"]
        rmax, omega = oracle.fora_setting(n, m, G["epsilon"], opt=case["opt"])
        assert (repr(rmax), repr(omega)) == (case["rmax"], case["omega"])
        rw, off, cnt = oracle.build_index(g, G["seed"], rmax, omega, opt=case["opt"])
        assert rw.size == case["index_total"] and sha(rw) == case["index_sha256"]
        for qd in case["queries"]:
            s = qd["source"]
            fifo = oracle.push_fifo(g, s, rmax)
            N, counts = oracle.walk_counts(fifo, omega, opt=case["opt"])
            assert repr(fifo["rsum"]) == qd["fifo"]["rsum"] and fifo["pops"] == qd["fifo"]["pops"]
            assert fifo["relax"] == qd["fifo"]["relax"] and N == qd["fifo"]["n_rw"] and int(counts.sum()) == qd["fifo"]["walks"]
            ppr, res, st = oracle.twin_query(g, s, rmax, omega, opt=case["opt"], seed=G["seed"])
            assert sha(ppr) == qd["twin"]["ppr_sha256"] and sha(res) == qd["twin"]["residue_sha256"]
            assert st["rsum_fix"] == qd["twin"]["rsum_fix"] and st["n_walks"] == qd["twin"]["n_walks"]


@pytest.mark.gpu
def test_hip_reproduces_golden(engine):
    G = _cases()
    for case in G["cases"]:
        n, m, (row_ptr, col) = _graph(case)
        engine.clear_index()
        engine.set_graph(n, m, row_ptr, col)
        engine.set_params(epsilon=G["epsilon"], opt=case["opt"], seed=G["seed"])
        rmax, omega = engine.get_params()
        assert (repr(rmax), repr(omega)) == (case["rmax"], case["omega"])
        srcs = np.array([qd["source"] for qd in case["queries"]], dtype=np.int32)
        ppr, res, st = engine.query_fix(srcs)
        for i, qd in enumerate(case["queries"]):
            t = qd["twin"]
            assert sha(ppr[i]) == t["ppr_sha256"] and sha(res[i]) == t["residue_sha256"]
            assert (st[i]["rsum_fix"], st[i]["levels"], st[i]["pops"], st[i]["relax"], st[i]["n_walks"]) == \
                (t["rsum_fix"], t["levels"], t["pops"], t["relax"], t["n_walks"])
        engine.build_index()
        rw, _, _ = engine.get_index()
        assert rw.size == case["index_total"] and sha(rw) == case["index_sha256"]
        ppr_i, _, _ = engine.query_fix(srcs, with_idx=True, want_residue=False)
        for i, qd in enumerate(case["queries"]):
            assert sha(ppr_i[i]) == qd["twin"]["ppr_idx_sha256"]
        engine.clear_index()
        if case["opt"]:
            ids, sc, rounds = engine.topk(srcs, 16, epsilon=G["epsilon"])
            for i, qd in enumerate(case["queries"]):
                assert ids[i].tolist() == qd["topk16"]["ids"] and rounds[i] == qd["topk16"]["rounds"]
                assert [float(x).hex() for x in sc[i]] == qd["topk16"]["scores_hex"]
