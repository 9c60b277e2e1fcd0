"""Synthetic graph / query-set generation for tests and bench.py.

The real edge lists of the BASELINE configs are not available
(/root/reference/.MISSING_LARGE_BLOBS:1), so every measured graph is an R-MAT
graph with the real graph's n and m.  Bench / test tooling only -- the query
hot path never touches this module.
"""
import numpy as np

# (n, m, seed) per SURVEY.md 8d
PRESETS = {
    "webstanford": (281_904, 2_312_497, 20260101),
    "livejournal": (4_847_571, 68_993_773, 20260102),
    "twitter2010": (41_652_230, 1_468_365_182, 20260103),
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


def query_set(n, count, seed):
    """`count` uniform source ids (the reference uses rand()%n, algo.h:498-509)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5351))
    return rng.integers(0, n, size=count, dtype=np.int64).astype(np.int32)


def preset(name, dangling="none"):
    n, m, seed = PRESETS[name]
    src, dst = rmat_graph(n, m, seed, dangling)
    row_ptr, col = csr_from_edges(n, src, dst)
    return n, m, row_ptr, col
