"""`fora` command line (fora_amd/bin/fora): flag handling and loader on CPU; build / query /
topk end to end on the GPU.  Mirrors fora.cpp:56-292 behaviour on the FORA path."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORA = os.path.join(ROOT, "fora_amd", "bin", "fora")


@pytest.fixture(scope="module")
def cli():
    import __graft_entry__
    __graft_entry__.build()
    assert os.path.exists(FORA)
    return FORA


def _write_dataset(folder, g, queries):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "attribute.txt"), "w") as f:
        f.write(f"n={g.n}\nm={g.m}\n")
    src = np.repeat(np.arange(g.n), np.diff(g.row_ptr))
    with open(os.path.join(folder, "graph.txt"), "w") as f:
        for s, d in zip(src.tolist(), g.col.tolist()):
            f.write(f"{s} {d}\n")
        f.write("7 7\n")  # a self loop the loader must drop (graph.h:157)
    with open(os.path.join(folder, "ssquery.txt"), "w") as f:
        for q in queries:
            f.write(f"{int(q)}\n")


def _run(args, **kw):
    return subprocess.run(args, capture_output=True, text=True, timeout=600, **kw)


def test_help_and_bad_flags(cli):
    r = _run([cli, "--help"])
    assert r.returncode == 0 and "fora query --algo <algo> [options]" in r.stdout
    r = _run([cli, "query", "--algo", "fora", "--nonsense"])
    assert r.returncode == 1 and "command not recognize --nonsense" in r.stderr   # fora.cpp:156-159
    r = _run([cli, "frobnicate"])
    assert r.returncode == 1 and "sub command not regoznized" in r.stderr         # fora.cpp:278-281
    r = _run([cli, "query", "--algo", "bippr", "--epsilon", "0.5"])
    assert r.returncode == 1
    r = _run([cli, "batch-topk", "--algo", "mc", "--epsilon", "0.5"])
    assert r.returncode == 1


def test_loader_matches_oracle(cli, oracle, tiny, tmp_path):
    folder = tmp_path / "data" / "toy"
    _write_dataset(str(folder), tiny, [1, 2, 3])
    r = _run([cli, "check-graph", "--prefix", str(tmp_path / "data") + "/", "--dataset", "toy"])
    assert r.returncode == 0, r.stderr
    assert f"init graph n: {tiny.n} m: {tiny.m}" in r.stdout
    g2 = oracle.Graph.from_folder(str(folder))
    assert (g2.row_ptr == tiny.row_ptr).all() and (g2.col == tiny.col).all()
    h = 1469598103934665603
    mask = (1 << 64) - 1
    for c in g2.col.tolist():
        h = ((h ^ (c & 0xFFFFFFFF)) * 1099511628211) & mask
    for p in g2.row_ptr.tolist():
        h = ((h ^ p) * 1099511628211) & mask
    assert f"nnz: {g2.col.size} csr_fnv1a: {h}" in r.stdout
    # missing graph file -> error exit like assert_file_exist (config.cpp:20-26)
    r = _run([cli, "check-graph", "--prefix", str(tmp_path / "data") + "/", "--dataset", "nope"])
    assert r.returncode == 1 and "not find" in r.stderr
    # id >= n -> assert in graph.h:155
    with open(folder / "graph.txt", "a") as f:
        f.write(f"0 {tiny.n}\n")
    r = _run([cli, "check-graph", "--prefix", str(tmp_path / "data") + "/", "--dataset", "toy"])
    assert r.returncode == 1


def test_query_without_gpu_fails_loudly(cli, tiny, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    folder = tmp_path / "data" / "toy"
    _write_dataset(str(folder), tiny, [1, 2, 3])
    r = _run([cli, "query", "--algo", "fora", "--prefix", str(tmp_path / "data") + "/", "--dataset", "toy",
              "--epsilon", "0.5"])
    assert r.returncode != 0 and "no usable MI355X" in r.stderr


@pytest.mark.gpu
def test_cli_build_query_topk_end_to_end(cli, oracle, small, tmp_path):
    from conftest import pick_sources
    g = small
    queries = pick_sources(g, 12, 41)
    folder = tmp_path / "data" / "g32k"
    _write_dataset(str(folder), g, queries)
    common = ["--prefix", str(tmp_path / "data") + "/", "--dataset", "g32k", "--epsilon", "0.5",
              "--result_dir", str(tmp_path / "res")]
    r = _run([cli, "build", *common])
    assert r.returncode == 0, r.stderr
    assert os.path.exists(folder / "randwalks.idx") and os.path.exists(folder / "randwalks.info")  # build.h:147-181
    rmax, omega = oracle.fora_setting(g.n, g.m, 0.5)
    total, _, _ = oracle.index_sizes(g, rmax, omega)
    assert f"tuned_index_size={total}" in r.stdout
    for extra in ([], ["--with_idx"]):
        r = _run([cli, "query", "--algo", "fora", "--query_size", "10", *common, *extra], env={**os.environ, "FORA_CLI_TIMING": "1"})
        assert r.returncode == 0, r.stderr
        assert "1. source node:%d" % queries[0] in r.stdout and "10. source node:%d" % queries[9] in r.stdout
        assert "11. source node" not in r.stdout                      # min(file, --query_size), query.h:1419
        assert "Total cost (s):" in r.stdout and "% for forward push cost" in r.stdout
        assert "Average query time (s):" in r.stdout and "Memory usage (MB):" in r.stdout
        if extra:
            assert "Average rand-walk idx hit ratio: 100%" in r.stdout
        name = "g32k.query.fora.%s.k-500.rmax-1.000000.json" % ("with_idx" if extra else "without_idx")
        j = json.load(open(tmp_path / "res" / "execution" / name))     # config.h:257-279
        assert j["config"]["algo"] == "fora" and j["config"]["query-size"] == "10"
        assert float(j["config"]["rmax"]) == rmax and float(j["config"]["omega"]) == omega
        assert int(j["result"]["n"]) == g.n and int(j["result"]["m"]) == g.m
        walks = sum(oracle.twin_query(g, int(s), rmax, omega, seed=0x464F5241)[2]["n_walks"] for s in queries[:10])
        assert float(j["result"]["total number of rand-walks"]) == walks
        assert set(j["timer"]) >= {"3", "5", "6"}
        # config.h:47-57 slots 5 (FWD_LU) / 6 (RONDOM_WALK) / 3 (FORA_QUERY) and the lines of algo.h:368-402 carry the
        # engine's real kernel times: the push slot covers every push kernel (round 4 left k_push_team out of it)
        et = dict(kv.split("=") for kv in next(l for l in r.stderr.splitlines() if l.startswith("engine_timing")).split()[1:])
        t3, t5, t6 = (float(j["timer"][k]) for k in ("3", "5", "6"))
        assert float(et["push_ms"]) > 0 and t5 >= 0.8 * float(et["push_ms"]) * 1e-3
        assert t6 >= 0.8 * float(et["walk_ms"]) * 1e-3
        assert t5 + t6 <= t3 * 1.001
        pct = {l.split("%")[1].strip(): float(l.split("%")[0]) for l in r.stdout.splitlines() if "% for " in l}
        assert pct["for forward push cost"] + pct["for random walk cost"] <= 100.0 + 1e-6
        assert abs(pct["for forward push cost"] - 100.0 * t5 / t3) < 0.5
    # the same index as Boost binary archives (build --boost_idx): same content, read back by query --with_idx
    h_native = _run([cli, "check-index", *common]).stdout.strip().split("\n")[-1]
    r = _run([cli, "build", "--boost_idx", *common])
    assert r.returncode == 0, r.stderr
    assert open(folder / "randwalks.idx", "rb").read(30)[8:] == b"serialization::archive"
    assert _run([cli, "check-index", *common]).stdout.strip().split("\n")[-1] == h_native
    r = _run([cli, "query", "--algo", "fora", "--query_size", "3", "--with_idx", *common])
    assert r.returncode == 0 and "Average rand-walk idx hit ratio: 100%" in r.stdout
    # --balanced (README.md:135 "TODS version"): adaptive rmax, same outputs; walk total equals the twin's
    r = _run([cli, "query", "--algo", "fora", "--balanced", "--query_size", "4", "--result_dir", str(tmp_path / "resb"), *common[:6]])
    assert r.returncode == 0, r.stderr
    jb = json.load(open(tmp_path / "resb" / "execution" / "g32k.query.fora.without_idx.k-500.rmax-1.000000.json"))
    walks_b = sum(oracle.twin_query_balanced(g, int(s), rmax, omega, seed=0x464F5241)[2]["n_walks"] for s in queries[:4])
    assert float(jb["result"]["total number of rand-walks"]) == walks_b
    # build --opt + topk --opt --with_idx
    r = _run([cli, "build", "--opt", *common])
    assert r.returncode == 0 and os.path.exists(folder / "randwalks.idx.onehopopt")
    r = _run([cli, "topk", "--algo", "fora", "--opt", "--with_idx", "--k", "20", "--query_size", "5", *common])
    assert r.returncode == 0, r.stderr
    assert "average iter times:" in r.stdout
    lines = open(tmp_path / "res" / "g32k.topk.k-20.txt").read().strip().split("\n")
    assert len(lines) == 5
    rmax_o, omega_o = oracle.fora_setting(g.n, g.m, 0.5, opt=True)
    index = oracle.build_index(g, 0x464F5241, rmax_o, omega_o, opt=True)
    ids, sc, _, _ = oracle.twin_topk_query(g, int(queries[0]), 20, 0.5, seed=0x464F5241, index=index)
    got = lines[0].split()
    assert int(got[0]) == queries[0]
    assert [int(x.split(":")[0]) for x in got[1:]] == ids.tolist()
    assert [float(x.split(":")[1]) for x in got[1:]] == sc.tolist()
    # --gpus 2: two host threads / contexts (both land on this box's single GPU), sources i mod 2, same results
    r2 = _run([cli, "topk", "--algo", "fora", "--opt", "--with_idx", "--k", "20", "--query_size", "5", "--gpus", "2", "--oversubscribe",
               "--result_dir", str(tmp_path / "res2"), *common[:6]])
    assert r2.returncode == 0, r2.stderr
    assert open(tmp_path / "res2" / "g32k.topk.k-20.txt").read() == open(tmp_path / "res" / "g32k.topk.k-20.txt").read()
    r2 = _run([cli, "query", "--algo", "fora", "--query_size", "10", "--gpus", "3", "--oversubscribe", "--result_dir", str(tmp_path / "res2"), *common[:6]])
    assert r2.returncode == 0, r2.stderr
    j2 = json.load(open(tmp_path / "res2" / "execution" / "g32k.query.fora.without_idx.k-500.rmax-1.000000.json"))
    j1 = json.load(open(tmp_path / "res" / "execution" / "g32k.query.fora.without_idx.k-500.rmax-1.000000.json"))
    assert j2["result"]["total number of rand-walks"] == j1["result"]["total number of rand-walks"]
    # topk without --opt is the bounds variant (get_topk, query.h:1150-1153), served from the non --opt index
    r = _run([cli, "topk", "--algo", "fora", "--with_idx", "--k", "20", "--query_size", "3", *common])
    assert r.returncode == 0, r.stderr
    lines = open(tmp_path / "res" / "g32k.topk.k-20.txt").read().strip().split("\n")
    index = oracle.build_index(g, 0x464F5241, rmax, omega, opt=False)
    ids, sc, _, _, _, _ = oracle.twin_topk_bound_query(g, int(queries[1]), 20, 0.5, seed=0x464F5241, index=index)
    got = lines[1].split()
    assert int(got[0]) == queries[1]
    assert [int(x.split(":")[0]) for x in got[1:]] == ids.tolist()
    assert [float(x.split(":")[1]) for x in got[1:]] == sc.tolist()


@pytest.mark.gpu
def test_cli_gen_exact_topk_and_precision(cli, oracle, small, tmp_path):
    """gen-exact-topk (query.h:1240-1307) writes the ground truth next to the graph; a later `topk` evaluates
    compute_precision (algo.h:524-572) against it and reports it like the reference (algo.h:394-398, config.h:227-228)."""
    from conftest import pick_sources
    g = small
    queries = pick_sources(g, 6, 43)
    folder = tmp_path / "data" / "g32k"
    _write_dataset(str(folder), g, queries)
    common = ["--prefix", str(tmp_path / "data") + "/", "--dataset", "g32k", "--result_dir", str(tmp_path / "res")]
    k = 20
    # topk before any ground truth exists: runs, says so (the reference would hit assert algo.h:395)
    r = _run([cli, "topk", "--algo", "fora", "--opt", "--epsilon", "0.5", "--k", str(k), "--query_size", "4", *common])
    assert r.returncode == 0, r.stderr
    assert "precision / recall not evaluated" in r.stdout
    r = _run([cli, "gen-exact-topk", "--k", str(k), "--query_size", "4", *common])
    assert r.returncode == 0, r.stderr
    assert "average generation time (s):" in r.stdout
    f = folder / "g32k.topk.pprs"                                       # build.h:121-125
    lines = open(f).read().strip().split("\n")
    assert lines[0] == "fora-exact-topk 1 4"
    want = {}
    for s in queries[:4]:
        fix, _ = oracle.twin_power_iteration(g, int(s), max_iter=100)
        nz = np.flatnonzero(fix)
        order = nz[np.lexsort((nz, -(fix[nz].astype(np.int64))))][:k]
        want[int(s)] = (order.tolist(), oracle.fix_to_double(fix[order]).tolist())
    for ln in lines[1:]:
        tok = ln.split()
        ids, sc = want[int(tok[0])]
        assert int(tok[1]) == k
        assert [int(x.split(":")[0]) for x in tok[2:]] == ids
        assert [float(x.split(":")[1]) for x in tok[2:]] == sc
    r2 = _run([cli, "gen-exact-topk", "--k", str(k), "--query_size", "4", *common])
    assert r2.returncode == 0 and "exact top k exists" in r2.stdout     # query.h:1251-1254
    r = _run([cli, "topk", "--algo", "fora", "--opt", "--epsilon", "0.5", "--k", str(k), "--query_size", "4", *common])
    assert r.returncode == 0, r.stderr
    assert "Average top-K Precision:" in r.stdout and "Average top-K Recall:" in r.stdout
    # recompute compute_precision from the two files
    got = open(tmp_path / "res" / f"g32k.topk.k-{k}.txt").read().strip().split("\n")
    prec = rec = 0.0
    for ln in got:
        tok = ln.split()
        est = {int(x.split(":")[0]) for x in tok[1:] if float(x.split(":")[1]) > 0}
        ex = set(want[int(tok[0])][0])
        rec += len(est & ex) / len(ex)
        prec += len(est & ex) / len(ex)
    j = json.load(open(tmp_path / "res" / "execution" / f"g32k.topk.fora.without_idx.k-{k}.rmax-1.000000.json"))
    assert abs(float(j["result"]["topk precision"]) - prec / 4) < 1e-12
    assert abs(float(j["result"]["topk recall"]) - rec / 4) < 1e-12
    assert float(j["result"]["topk precision"]) >= 0.8
    # batch-topk (query.h:1517-1640): the algorithm again for k/5, 2k/5, ..., k; precision table at the end
    r = _run([cli, "batch-topk", "--algo", "fora", "--opt", "--epsilon", "0.5", "--k", str(k), "--query_size", "4", *common])
    assert r.returncode == 0, r.stderr
    for kk in (4, 8, 12, 16, 20):
        assert f"k is set to be ={kk}" in r.stdout and f"k={kk} precision=" in r.stdout
    tail = r.stdout.strip().split("\n")[-6:]
    assert tail[0] == "fora" and tail[1].split() == ["4", "8", "12", "16", "20"] and tail[2] == "Precision:" and tail[4] == "Recall:"
    assert abs(float(tail[3].split()[-1]) - prec / 4) < 1e-5            # the k = 20 column is the topk run above


def _fnv(rw, off, cnt):
    h = 1469598103934665603
    M = (1 << 64) - 1
    for v in rw:
        h = ((h ^ (int(v) & 0xFFFFFFFF)) * 1099511628211) & M
    for o, c in zip(off, cnt):
        h = ((h ^ int(o)) * 1099511628211) & M
        h = ((h ^ int(c)) * 1099511628211) & M
    return h


def test_index_files_boost_archive_reader(cli, tmp_path):
    """deserialize_idx (build.h:194-207) reads Boost binary archives.  No Boost here, so the layout is the presumed one
    (unpinned); the reader anchors on the signature, the payload at the END of the file and the count in front of it,
    so extra preamble bytes (class info, item version) do not matter.  `fora check-index` needs no GPU."""
    import struct
    n = 50
    rng = np.random.Generator(np.random.PCG64(5))
    cnt = rng.integers(0, 7, size=n).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.uint64)
    rw = rng.integers(0, n, size=int(cnt.sum())).astype(np.int32)
    folder = tmp_path / "data" / "g"
    os.makedirs(folder)
    open(folder / "attribute.txt", "w").write(f"n={n}\nm=100\n")
    header = struct.pack("<Q", 22) + b"serialization::archive" + struct.pack("<H", 15) + bytes([4, 8, 4, 8]) + struct.pack("<I", 1)
    pairs = b"".join(struct.pack("<QQ", int(o), int(c)) for o, c in zip(off, cnt))
    want = f"index walks: {rw.size} nodes: {n} fnv1a: {_fnv(rw, off, cnt)}"
    common = ["--prefix", str(tmp_path / "data") + "/", "--dataset", "g"]
    for preamble in (struct.pack("<hBI", 0, 0, 0), b"", struct.pack("<hBIxxxx", 0, 0, 0)):
        open(folder / "randwalks.idx", "wb").write(header + struct.pack("<Q", rw.size) + rw.tobytes())
        open(folder / "randwalks.info", "wb").write(header + preamble + struct.pack("<Q", n) + pairs)
        r = _run([cli, "check-index", *common])
        assert r.returncode == 0, r.stderr
        assert want in r.stdout
    # corrupt count / truncated payload / foreign file are refused with a reason
    open(folder / "randwalks.info", "wb").write(header + struct.pack("<Q", n + 1) + pairs)
    r = _run([cli, "check-index", *common])
    assert r.returncode == 1 and "element count" in r.stderr
    open(folder / "randwalks.info", "wb").write(header + struct.pack("<Q", n) + pairs)
    open(folder / "randwalks.idx", "wb").write(header + struct.pack("<Q", rw.size) + rw.tobytes()[:-4])
    r = _run([cli, "check-index", *common])
    assert r.returncode == 1
    open(folder / "randwalks.idx", "wb").write(b"not an archive at all, just bytes " * 4)
    r = _run([cli, "check-index", *common])
    assert r.returncode == 1
