"""ctypes binding of oracle/libfora_oracle.so -- the CPU checker.

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never by fora_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

FIX_ONE = 1 << 62
STREAM_INDEX = 0xFFFFFFFF


class PushStats(C.Structure):
    _fields_ = [("rsum", C.c_double), ("pops", C.c_int64), ("relax", C.c_int64),
                ("n_reserve", C.c_int64), ("n_residue", C.c_int64), ("generations", C.c_int64)]


class RefineStats(C.Structure):
    _fields_ = [("n_walks", C.c_uint64), ("n_idx_hit", C.c_uint64), ("walk_steps", C.c_uint64)]


class TwinPushStats(C.Structure):
    _fields_ = [("rsum_fix", C.c_uint64), ("levels", C.c_int64), ("pops", C.c_int64),
                ("relax", C.c_int64)]


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libfora_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_walk.restype = C.c_int32
        _LIB.orc_build_csr.restype = C.c_int64
        _LIB.orc_count_edges.restype = C.c_int64
        _LIB.orc_read_queries.restype = C.c_int64
        _LIB.orc_walk_counts.restype = C.c_uint64
        _LIB.orc_index_sizes.restype = C.c_uint64
        _LIB.orc_twin_alpha_fix.restype = C.c_uint64
        _LIB.orc_twin_rmax_fix.restype = C.c_uint64
        _LIB.orc_twin_walk_counts.restype = C.c_uint64
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _d(x):
    return C.c_double(float(x))


class Graph:
    """CSR graph as the oracle sees it."""

    def __init__(self, n, m_attr, row_ptr, col):
        self.n = int(n)
        self.m = int(m_attr)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        assert self.row_ptr.shape == (self.n + 1,)

    @property
    def deg(self):
        return np.diff(self.row_ptr)

    @classmethod
    def from_edges(cls, n, m_attr, src, dst):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        col = np.zeros(max(1, src.size), dtype=np.int32)
        nnz = lib().orc_build_csr(C.c_int32(n), _p(src), _p(dst), C.c_int64(src.size), _p(row_ptr), _p(col))
        if nnz < 0:
            raise ValueError("node id out of range")
        return cls(n, m_attr, row_ptr, col[:nnz])

    @classmethod
    def from_folder(cls, folder):
        n, m = read_attribute(os.path.join(folder, "attribute.txt"))
        gpath = os.path.join(folder, "graph.txt").encode()
        ne = lib().orc_count_edges(gpath)
        if ne < 0:
            raise FileNotFoundError(gpath)
        src = np.zeros(max(1, ne), dtype=np.int32)
        dst = np.zeros(max(1, ne), dtype=np.int32)
        got = C.c_int64(0)
        rc = lib().orc_read_edges(gpath, _p(src), _p(dst), C.c_int64(ne), C.byref(got))
        assert rc == 0
        return cls.from_edges(n, m, src[:got.value], dst[:got.value])


def read_attribute(path):
    n, m = C.c_int32(0), C.c_int64(0)
    rc = lib().orc_read_attribute(path.encode(), C.byref(n), C.byref(m))
    if rc:
        raise IOError(f"attribute file {path}: rc={rc}")
    return n.value, m.value


def read_queries(path, cap=1 << 20):
    out = np.zeros(cap, dtype=np.int32)
    k = lib().orc_read_queries(path.encode(), _p(out), C.c_int64(cap))
    if k < 0:
        raise FileNotFoundError(path)
    return out[:k].copy()


def fora_setting(n, m, epsilon, alpha=0.2, rmax_scale=1.0, opt=False):
    rmax, omega = C.c_double(0), C.c_double(0)
    lib().orc_fora_setting(C.c_int32(n), C.c_int64(m), _d(epsilon), _d(alpha), _d(rmax_scale),
                           C.c_int(int(opt)), C.byref(rmax), C.byref(omega))
    return rmax.value, omega.value


def fora_topk_setting(m, epsilon, delta, pfail, rmax_scale=1.0):
    rmax, omega = C.c_double(0), C.c_double(0)
    lib().orc_fora_topk_setting(C.c_int64(m), _d(epsilon), _d(delta), _d(pfail), _d(rmax_scale),
                                C.byref(rmax), C.byref(omega))
    return rmax.value, omega.value


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def walk(g, seed, stream, rnd, start, j, alpha=0.2, no_zero_hop=False):
    return lib().orc_walk(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_uint64(seed), C.c_uint32(stream),
                          C.c_uint32(rnd), C.c_int32(start), C.c_uint64(j), _d(alpha),
                          C.c_int(int(no_zero_hop)), None)


def walk_steps(g, seed, stream, rnd, start, j, alpha=0.2, no_zero_hop=False):
    steps = C.c_int64(0)
    lib().orc_walk(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_uint64(seed), C.c_uint32(stream),
                   C.c_uint32(rnd), C.c_int32(start), C.c_uint64(j), _d(alpha), C.c_int(int(no_zero_hop)), C.byref(steps))
    return steps.value


def push_fifo(g, s, rmax, alpha=0.2):
    reserve = np.zeros(g.n, dtype=np.float64)
    residue = np.zeros(g.n, dtype=np.float64)
    o1 = np.zeros(g.n, dtype=np.int32)
    o2 = np.zeros(g.n, dtype=np.int32)
    st = PushStats()
    lib().orc_push_fifo(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _d(rmax), _d(alpha),
                        _p(reserve), _p(residue), _p(o1), _p(o2), C.byref(st))
    return dict(reserve=reserve, residue=residue, reserve_occur=o1[:st.n_reserve].copy(),
                residue_occur=o2[:st.n_residue].copy(), rsum=st.rsum, pops=st.pops, relax=st.relax,
                generations=st.generations)


def walk_counts(push, omega, alpha=0.2, opt=False):
    occ = np.ascontiguousarray(push["residue_occur"], dtype=np.int32)
    out = np.zeros(max(1, occ.size), dtype=np.uint64)
    N = lib().orc_walk_counts(_p(push["residue"]), _p(occ), C.c_int64(occ.size), _d(push["rsum"]),
                              _d(omega), _d(alpha), C.c_int(int(opt)), _p(out))
    return N, out[:occ.size]


def index_sizes(g, rmax, omega, alpha=0.2, opt=False):
    off = np.zeros(g.n, dtype=np.uint64)
    cnt = np.zeros(g.n, dtype=np.uint64)
    total = lib().orc_index_sizes(C.c_int32(g.n), _p(g.row_ptr), _d(rmax), _d(omega), _d(alpha),
                                  C.c_int(int(opt)), _p(off), _p(cnt))
    return total, off, cnt


def build_index(g, seed, rmax, omega, alpha=0.2, opt=False):
    total, off, cnt = index_sizes(g, rmax, omega, alpha, opt)
    rw = np.zeros(max(1, total), dtype=np.int32)
    lib().orc_build_index(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_uint64(seed), _d(alpha),
                          C.c_int(int(opt)), _p(off), _p(cnt), _p(rw))
    return rw[:total], off, cnt


def _idx_args(index):
    if index is None:
        return None, None, None
    rw, off, cnt = index
    return _p(rw), _p(off), _p(cnt)


def refine(g, s, push, omega, alpha=0.2, opt=False, seed=0, index=None):
    ppr = np.zeros(g.n, dtype=np.float64)
    st = RefineStats()
    occ2 = np.ascontiguousarray(push["residue_occur"], dtype=np.int32).copy()
    occ1 = np.ascontiguousarray(push["reserve_occur"], dtype=np.int32)
    a, b, c = _idx_args(index)
    lib().orc_refine(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _p(push["reserve"]), _p(occ1),
                     C.c_int64(occ1.size), _p(push["residue"]), _p(occ2), C.c_int64(occ2.size),
                     _d(push["rsum"]), _d(omega), _d(alpha), C.c_int(int(opt)), C.c_uint64(seed), a, b, c,
                     _p(ppr), C.byref(st))
    return ppr, dict(n_walks=st.n_walks, n_idx_hit=st.n_idx_hit, walk_steps=st.walk_steps)


def query(g, s, rmax, omega, alpha=0.2, opt=False, seed=0, index=None):
    ppr = np.zeros(g.n, dtype=np.float64)
    ps, rs = PushStats(), RefineStats()
    a, b, c = _idx_args(index)
    lib().orc_query(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _d(rmax), _d(omega), _d(alpha),
                    C.c_int(int(opt)), C.c_uint64(seed), a, b, c, _p(ppr), C.byref(ps), C.byref(rs))
    return ppr, dict(rsum=ps.rsum, pops=ps.pops, relax=ps.relax, n_walks=rs.n_walks,
                     n_idx_hit=rs.n_idx_hit, walk_steps=rs.walk_steps)


def query_many(g, sources, rmax, omega, threads, seconds, alpha=0.2, opt=False, seed=0, index=None, cpus=None):
    """The query loop on `threads` host threads (pthreads inside the oracle, private buffers per thread) for at most
    `seconds`: (queries finished, wall seconds, walks).  cpus: thread t binds itself to cpus[t mod len(cpus)]."""
    src = np.ascontiguousarray(sources, dtype=np.int32)
    a, b, c = _idx_args(index)
    el, walks = C.c_double(0), C.c_uint64(0)
    f = lib().orc_query_many_pinned
    f.restype = C.c_int64
    pin = np.ascontiguousarray(cpus, dtype=np.int32) if cpus is not None and len(cpus) else None
    done = f(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), _p(src), C.c_int64(src.size), _d(rmax), _d(omega), _d(alpha),
             C.c_int(int(opt)), C.c_uint64(seed), a, b, c, C.c_int(int(threads)), _d(seconds), C.byref(el), C.byref(walks),
             _p(pin), C.c_int(0 if pin is None else int(pin.size)))
    return int(done), float(el.value), int(walks.value)


def topk_query(g, s, k, epsilon, alpha=0.2, rmax_scale=1.0, seed=0, index=None, want_ppr=False):
    ids = np.zeros(k, dtype=np.int32)
    sc = np.zeros(k, dtype=np.float64)
    rounds = C.c_int32(0)
    ppr = np.zeros(g.n, dtype=np.float64) if want_ppr else None
    a, b, c = _idx_args(index)
    lib().orc_topk_query(C.c_int32(g.n), C.c_int64(g.m), _p(g.row_ptr), _p(g.col), C.c_int32(s), C.c_int32(k),
                         _d(epsilon), _d(alpha), _d(rmax_scale), C.c_uint64(seed), a, b, c, _p(ids), _p(sc),
                         C.byref(rounds), _p(ppr))
    return ids, sc, rounds.value, ppr


def topk_push_counts(g, s, k, epsilon, rounds, alpha=0.2, rmax_scale=1.0):
    """(pops, relaxations) of the FIFO pushes of the first `rounds` rounds of the --opt top-k driver (no walks)."""
    pops, relax = C.c_int64(0), C.c_int64(0)
    lib().orc_topk_push_counts(C.c_int32(g.n), C.c_int64(g.m), _p(g.row_ptr), _p(g.col), C.c_int32(s), C.c_int32(k),
                               _d(epsilon), _d(alpha), _d(rmax_scale), C.c_int32(int(rounds)), C.byref(pops), C.byref(relax))
    return int(pops.value), int(relax.value)


def power_iteration(g, s, alpha=0.2, iters=100):
    ppr = np.zeros(g.n, dtype=np.float64)
    lib().orc_power_iteration(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _d(alpha),
                              C.c_int(iters), _p(ppr))
    return ppr


# ------------------------------------------------------------------ twin
def fix_to_double(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros(a.shape, dtype=np.float64)
    lib().orc_fix_to_double(_p(a), C.c_int64(a.size), _p(out))
    return out


def twin_push(g, s, rmax, alpha=0.2, max_levels=1 << 16):
    residue = np.zeros(g.n, dtype=np.uint64)
    ppr = np.zeros(g.n, dtype=np.uint64)
    lv = np.zeros(max_levels, dtype=np.int64)
    st = TwinPushStats()
    lib().orc_twin_push(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _d(rmax), _d(alpha), _p(residue),
                        _p(ppr), C.byref(st), _p(lv), C.c_int64(max_levels))
    return dict(residue=residue, reserve=ppr, rsum_fix=st.rsum_fix, levels=st.levels, pops=st.pops,
                relax=st.relax, level_sizes=lv[:min(st.levels, max_levels)].copy())


def twin_power_iteration(g, s, max_iter=100, alpha=0.2):
    ppr = np.zeros(g.n, dtype=np.uint64)
    st = TwinPushStats()
    lib().orc_twin_power_iteration(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _d(alpha),
                                   C.c_int32(max_iter), _p(ppr), C.byref(st))
    return ppr, dict(rsum_fix=st.rsum_fix, levels=st.levels, pops=st.pops, relax=st.relax)


def twin_walk_counts(g, residue, rsum_fix, omega, alpha=0.2, opt=False):
    out = np.zeros(g.n, dtype=np.uint64)
    N = lib().orc_twin_walk_counts(C.c_int32(g.n), _p(residue), C.c_uint64(rsum_fix), _d(omega), _d(alpha),
                                   C.c_int(int(opt)), _p(out))
    return N, out


def twin_set_defer(k):
    """Bounded deferral of the twin's push (the engine's option \"defer\"; default 0: plain levels)."""
    lib().orc_twin_set_defer(C.c_int(int(k)))


def twin_set_defer_min(m):
    lib().orc_twin_set_defer_min(C.c_int64(int(m)))


def twin_get_defer():
    return int(lib().orc_twin_get_defer())


def twin_set_rounds(rounds):
    """Threshold rounds of the twin's push (the engine's option \"rounds\", default 1)."""
    lib().orc_twin_set_rounds(C.c_int(int(rounds)))


def twin_set_round_div(div):
    """Leave a threshold round once the frontier is down to 1/div of the round's largest (0: only when it is empty)."""
    lib().orc_twin_set_round_div(C.c_int(int(div)))


def twin_query(g, s, rmax, omega, alpha=0.2, opt=False, seed=0, index=None):
    residue = np.zeros(g.n, dtype=np.uint64)
    ppr = np.zeros(g.n, dtype=np.uint64)
    ps, rs = TwinPushStats(), RefineStats()
    a, b, c = _idx_args(index)
    lib().orc_twin_query(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _d(rmax), _d(omega), _d(alpha),
                         C.c_int(int(opt)), C.c_uint64(seed), a, b, c, _p(residue), _p(ppr), C.byref(ps),
                         C.byref(rs))
    return ppr, residue, dict(rsum_fix=ps.rsum_fix, levels=ps.levels, pops=ps.pops, relax=ps.relax,
                              n_walks=rs.n_walks, n_idx_hit=rs.n_idx_hit, walk_steps=rs.walk_steps)


def twin_topk_query(g, s, k, epsilon, alpha=0.2, rmax_scale=1.0, seed=0, index=None, want_ppr=False):
    ids = np.zeros(k, dtype=np.int32)
    sc = np.zeros(k, dtype=np.float64)
    rounds = C.c_int32(0)
    ppr = np.zeros(g.n, dtype=np.uint64) if want_ppr else None
    a, b, c = _idx_args(index)
    lib().orc_twin_topk_query(C.c_int32(g.n), C.c_int64(g.m), _p(g.row_ptr), _p(g.col), C.c_int32(s),
                              C.c_int32(k), _d(epsilon), _d(alpha), _d(rmax_scale), C.c_uint64(seed), a, b, c,
                              _p(ids), _p(sc), C.byref(rounds), _p(ppr))
    return ids, sc, rounds.value, ppr


def topk_bound_query(g, s, k, epsilon, alpha=0.2, rmax_scale=1.0, ppr_decay_alpha=0.77, seed=0, index=None,
                     want_ppr=False):
    """get_topk without --opt (fora_query_topk_with_bound, query.h:909-969), reference arithmetic."""
    ids = np.zeros(k, dtype=np.int32)
    sc = np.zeros(k, dtype=np.float64)
    rounds = C.c_int32(0)
    ppr = np.zeros(g.n, dtype=np.float64) if want_ppr else None
    a, b, c = _idx_args(index)
    lib().orc_topk_bound_query(C.c_int32(g.n), C.c_int64(g.m), _p(g.row_ptr), _p(g.col), C.c_int32(s), C.c_int32(k),
                               _d(epsilon), _d(alpha), _d(rmax_scale), _d(ppr_decay_alpha), C.c_uint64(seed), a, b, c,
                               _p(ids), _p(sc), C.byref(rounds), _p(ppr))
    return ids, sc, rounds.value, ppr


def twin_topk_bound_query(g, s, k, epsilon, alpha=0.2, rmax_scale=1.0, ppr_decay_alpha=0.77, seed=0, index=None,
                          want_ppr=False, want_bounds=False):
    ids = np.zeros(k, dtype=np.int32)
    sc = np.zeros(k, dtype=np.float64)
    rounds = C.c_int32(0)
    ppr = np.zeros(g.n, dtype=np.uint64) if want_ppr else None
    up = np.zeros(g.n, dtype=np.float64) if want_bounds else None
    lo = np.zeros(g.n, dtype=np.float64) if want_bounds else None
    a, b, c = _idx_args(index)
    lib().orc_twin_topk_bound_query(C.c_int32(g.n), C.c_int64(g.m), _p(g.row_ptr), _p(g.col), C.c_int32(s),
                                    C.c_int32(k), _d(epsilon), _d(alpha), _d(rmax_scale), _d(ppr_decay_alpha),
                                    C.c_uint64(seed), a, b, c, _p(ids), _p(sc), C.byref(rounds), _p(ppr), _p(up), _p(lo))
    return ids, sc, rounds.value, ppr, up, lo


def calculate_lambda(rsum, pfail, upper_bound, total_rw_num):
    f = lib().orc_calculate_lambda
    f.restype = C.c_double
    return f(_d(rsum), _d(pfail), _d(upper_bound), C.c_long(total_rw_num))


# MI355X cost model of --balanced (seconds): measured on the headline workload
BAL_C_POP, BAL_C_EDGE, BAL_T_WALK, BAL_T_IDX = 2.0e-11, 2.4e-11, 6.5e-11, 2.2e-11


def twin_query_balanced(g, s, rmax, omega, alpha=0.2, opt=False, seed=0, index=None, start_scale=0.0, c_pop=BAL_C_POP,
                        c_edge=BAL_C_EDGE, t_walk=BAL_T_WALK, t_idx=BAL_T_IDX):
    residue = np.zeros(g.n, dtype=np.uint64)
    ppr = np.zeros(g.n, dtype=np.uint64)
    ps, rs = TwinPushStats(), RefineStats()
    rm = C.c_double(0)
    a, b, c = _idx_args(index)
    f = lib().orc_twin_query_balanced
    rounds = f(C.c_int32(g.n), _p(g.row_ptr), _p(g.col), C.c_int32(s), _d(rmax), _d(omega), _d(alpha),
               C.c_int(int(opt)), C.c_uint64(seed), a, b, c, _d(start_scale), _d(c_pop), _d(c_edge), _d(t_walk), _d(t_idx),
               _p(residue), _p(ppr), C.byref(ps), C.byref(rs), C.byref(rm))
    return ppr, residue, dict(rsum_fix=ps.rsum_fix, levels=ps.levels, pops=ps.pops, relax=ps.relax,
                              n_walks=rs.n_walks, n_idx_hit=rs.n_idx_hit, rounds=rounds, rmax=rm.value)
