"""Synthetic graph / query-set generation for tests and bench.py.

The real edge lists of the BASELINE configs are not available
(/root/reference/.MISSING_LARGE_BLOBS:1), so every measured graph is an R-MAT
graph with the real graph's n and m.  Bench / test tooling only -- the query
hot path never touches this module.
"""
import ctypes
import os

import numpy as np

# (n, m, seed) per SURVEY.md 8d
PRESETS = {
    "webstanford": (281_904, 2_312_497, 20260101),
    "livejournal": (4_847_571, 68_993_773, 20260102),
    "twitter2010": (41_652_230, 1_468_365_182, 20260103),
    "medium": (1_500_000, 16_000_000, 20260104),  # 184 bins: the wide bucket layout at a size the CPU twin checks in seconds
    # scaled-down ladder for CI
    "tiny": (2_000, 16_000, 20260111),
    "small": (32_000, 262_000, 20260112),
}


def _rmat_pairs(rng, scale, count, a=0.57, b=0.19, c=0.19):
    src = np.zeros(count, dtype=np.int64)
    dst = np.zeros(count, dtype=np.int64)
    ab, abc = a + b, a + b + c
    for _ in range(scale):
        r = rng.random(count)
        src = (src << 1) | (r >= ab)
        dst = (dst << 1) | (((r >= a) & (r < ab)) | (r >= abc))
    return src, dst


def rmat_graph(n, m, seed, dangling="none", fold=False):
    """Directed R-MAT (0.57, 0.19, 0.19, 0.05) edge list with exactly m distinct
    non-loop edges over n nodes, node ids randomly permuted.

    dangling="none": every node first gets one uniform random out-edge (real
      web-Stanford has almost no zero-out-degree nodes), the rest is R-MAT.
    dangling="rmat": plain R-MAT (about 43 % zero-out-degree nodes at ws size).
    fold=True maps R-MAT ids >= n back with `% n` instead of rejecting the pair (3-4x fewer samples when n is
    far from a power of two; slightly different degree law) -- only for the one-off large-scale runs, the
    test / bench graphs keep fold=False.
    Returns (src, dst) int32 arrays sorted by (src, dst).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    scale = max(1, int(np.ceil(np.log2(n))))
    perm = rng.permutation(n).astype(np.int64)
    keys = np.empty(0, dtype=np.int64)
    if dangling == "none":
        u = np.arange(n, dtype=np.int64)
        v = rng.integers(0, n - 1, size=n, dtype=np.int64)
        v = v + (v >= u)  # uniform over nodes != u
        keys = np.unique(u * n + v)
    elif dangling != "rmat":
        raise ValueError(dangling)
    if keys.size > m:
        raise ValueError("m < n: cannot give every node an out-edge")
    base = keys
    extra = np.empty(0, dtype=np.int64)
    need = m - base.size
    while extra.size < need:
        want = int((need - extra.size) * 1.3) + 1024
        s, d = _rmat_pairs(rng, scale, want)
        if fold:
            s %= n
            d %= n
        ok = (s < n) & (d < n) & (s != d)
        k = perm[s[ok]] * n + perm[d[ok]]
        # keep first occurrences in generation order so truncation is unbiased
        k = k[np.sort(np.unique(k, return_index=True)[1])]
        k = k[~np.isin(k, base)]
        if extra.size:
            k = k[~np.isin(k, extra)]
        extra = np.concatenate([extra, k])
    keys = np.sort(np.concatenate([base, extra[:need]]))
    return (keys // n).astype(np.int32), (keys % n).astype(np.int32)


def csr_from_edges(n, src, dst):
    """CSR keeping per-row input order; self loops dropped, duplicates kept
    (loader semantics of /root/reference/graph.h:151-161)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    order = np.argsort(src, kind="stable")
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=row_ptr[1:])
    return row_ptr, dst[order].astype(np.int32)


# Graphs of these presets are built by the HIP generator tools/rmat_gen.hip (a Twitter-2010-sized graph takes the
# numpy generator above about 13 minutes and 60 GB; the GPU needs seconds) -- they need a GPU, like everything that
# uses them.  The small presets keep the numpy generator, so that CPU-only tests and the committed golden vectors
# see the same graphs as before.
GPU_PRESETS = ("medium", "livejournal", "twitter2010")
_RMAT_LIB = None


def rmat_csr_gpu(n, m, seed, dangling="none"):
    """(row_ptr int64[n+1], col int32[m]) of the HIP generator: same model as rmat_graph (R-MAT 0.57/0.19/0.19/0.05,
    ids permuted, no loops, no duplicates, exactly m edges, dangling="none" gives every node one uniform out-edge
    first), different random stream.  Deterministic in (n, m, seed, dangling).  Rows come out sorted by target."""
    global _RMAT_LIB
    if _RMAT_LIB is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "librmat_gen.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _RMAT_LIB = ctypes.CDLL(path)
    if dangling not in ("none", "rmat"):
        raise ValueError(dangling)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    col = np.zeros(max(1, m), dtype=np.int32)
    secs = ctypes.c_double(0)
    rc = _RMAT_LIB.rmat_generate(ctypes.c_int64(n), ctypes.c_int64(m), ctypes.c_uint64(seed), ctypes.c_int(dangling == "rmat"),
                                 row_ptr.ctypes.data_as(ctypes.c_void_p), col.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.byref(secs))
    if rc:
        raise RuntimeError(f"rmat_generate failed: rc={rc} (no GPU, or not enough memory?)")
    return row_ptr, col[:m]


def query_set(n, count, seed):
    """`count` uniform source ids (the reference uses rand()%n, algo.h:498-509)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5351))
    return rng.integers(0, n, size=count, dtype=np.int64).astype(np.int32)


def preset(name, dangling="none"):
    n, m, seed = PRESETS[name]
    if name in GPU_PRESETS:
        row_ptr, col = rmat_csr_gpu(n, m, seed, dangling)
        return n, m, row_ptr, col
    src, dst = rmat_graph(n, m, seed, dangling)
    row_ptr, col = csr_from_edges(n, src, dst)
    return n, m, row_ptr, col
