"""ctypes binding of libfora_hip.so (include/fora_hip.h).  Thin: every method is
one C-ABI call; errors raise ForaError with fora_hip_last_error()."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
FIX_ONE = 1 << 62
STREAM_INDEX = 0xFFFFFFFF

SYMBOLS = [
    "fora_hip_create", "fora_hip_destroy", "fora_hip_device_count", "fora_hip_last_error", "fora_hip_device_info",
    "fora_hip_set_graph", "fora_hip_set_params", "fora_hip_set_params_raw", "fora_hip_get_params",
    "fora_hip_set_batch", "fora_hip_get_batch", "fora_hip_set_option", "fora_hip_get_option", "fora_hip_set_balanced", "fora_hip_index_sizes", "fora_hip_build_index",
    "fora_hip_get_index", "fora_hip_set_index", "fora_hip_clear_index", "fora_hip_query_batch",
    "fora_hip_query_batch_fix", "fora_hip_topk_batch", "fora_hip_topk_bound_batch", "fora_hip_power_iteration_batch", "fora_hip_push_batch", "fora_hip_walk_counts",
    "fora_hip_walks", "fora_hip_reset_timing", "fora_hip_get_timing", "fora_hip_get_stamps",
]


class ForaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fora_hip error {code}: {msg}")
        self.code = code


class QueryStats(C.Structure):
    _fields_ = [("rsum", C.c_double), ("rsum_fix", C.c_uint64), ("n_rw", C.c_uint64),
                ("n_walks", C.c_uint64), ("n_idx_hit", C.c_uint64), ("pops", C.c_uint64),
                ("relax", C.c_uint64), ("ppr_sum_fix", C.c_uint64), ("levels", C.c_int32),
                ("dangling_source", C.c_int32), ("rmax_used", C.c_double), ("push_rounds", C.c_int32),
                ("reserved_", C.c_int32)]


_STATS_DTYPE = np.dtype([(name, {C.c_double: np.float64, C.c_uint64: np.uint64, C.c_int32: np.int32}[ct])
                         for name, ct in QueryStats._fields_], align=True)
assert _STATS_DTYPE.itemsize == C.sizeof(QueryStats)


class Timing(C.Structure):
    _fields_ = [("push_pop_ms", C.c_double), ("push_expand_ms", C.c_double), ("push_accum_ms", C.c_double), ("walk_alloc_ms", C.c_double),
                ("walk_ms", C.c_double), ("walk_accum_ms", C.c_double), ("other_ms", C.c_double), ("batch_ms", C.c_double),
                ("push_pop_launches", C.c_uint64), ("push_expand_launches", C.c_uint64), ("push_accum_launches", C.c_uint64),
                ("walk_launches", C.c_uint64), ("batches", C.c_uint64), ("pops", C.c_uint64),
                ("relax", C.c_uint64), ("walks", C.c_uint64), ("walk_steps", C.c_uint64),
                ("levels", C.c_uint64), ("idx_hits", C.c_uint64), ("push_tail_ms", C.c_double), ("push_tail_launches", C.c_uint64),
                ("push_team_ms", C.c_double), ("push_team_launches", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib_path():
    # FORA_HIP_LIB: experiment tooling only (tools/pushbench.py times several builds of the library)
    return os.environ.get("FORA_HIP_LIB") or os.path.join(_HERE, "libfora_hip.so")


_OTHER = {}


def load(path=None):
    """Loads the in-tree HIP library (path: another build of it -- libfora_hip_test.so, the build with the schedule
    experiments compiled in, for their twin-equivalence tests).  Fails loudly when it has not been built."""
    global _LIB
    if path is not None:
        if path not in _OTHER:
            if not os.path.exists(path):
                raise ImportError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
            lib = C.CDLL(path)
            lib.fora_hip_last_error.restype = C.c_char_p
            lib.fora_hip_destroy.restype = None
            _OTHER[path] = lib
        return _OTHER[path]
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError(f"{p} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = C.CDLL(p)
        _LIB.fora_hip_last_error.restype = C.c_char_p
        _LIB.fora_hip_destroy.restype = None
    return _LIB


TEST_LIB = os.path.join(_HERE, "libfora_hip_test.so")  # -DFORA_TEST_PATHS=1: threshold rounds / bounded deferral compiled in


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """One GPU context (fora_ctx)."""

    def __init__(self, device=0, lib=None):
        self._lib = load(lib)
        self._ctx = C.c_void_p()
        rc = self._lib.fora_hip_create(C.c_int(device), C.byref(self._ctx))
        if rc:
            raise ForaError(rc, "fora_hip_create failed (no gfx950 GPU visible?)")
        self.n = 0

    def close(self):
        if self._ctx:
            self._lib.fora_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise ForaError(rc, (self._lib.fora_hip_last_error(self._ctx) or b"").decode())

    def device_info(self):
        arch = C.create_string_buffer(64)
        cus, hbm = C.c_int(0), C.c_uint64(0)
        self._chk(self._lib.fora_hip_device_info(self._ctx, arch, C.c_int(64), C.byref(cus), C.byref(hbm)))
        return arch.value.decode(), cus.value, hbm.value

    def set_graph(self, n, m_attr, row_ptr, col):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        self._chk(self._lib.fora_hip_set_graph(self._ctx, C.c_int32(n), C.c_int64(m_attr), _p(row_ptr), _p(col)))
        self.n = int(n)

    def set_params(self, alpha=0.2, epsilon=0.5, rmax_scale=1.0, opt=False, seed=0):
        self._chk(self._lib.fora_hip_set_params(self._ctx, C.c_double(alpha), C.c_double(epsilon),
                                                C.c_double(rmax_scale), C.c_int(int(opt)), C.c_uint64(seed)))

    def set_params_raw(self, alpha, rmax, omega, opt=False, seed=0):
        self._chk(self._lib.fora_hip_set_params_raw(self._ctx, C.c_double(alpha), C.c_double(rmax),
                                                    C.c_double(omega), C.c_int(int(opt)), C.c_uint64(seed)))

    def get_params(self):
        rmax, omega = C.c_double(0), C.c_double(0)
        self._chk(self._lib.fora_hip_get_params(self._ctx, C.byref(rmax), C.byref(omega)))
        return rmax.value, omega.value

    def set_batch(self, batch):
        self._chk(self._lib.fora_hip_set_batch(self._ctx, C.c_int(batch)))

    def get_batch(self):
        return self._lib.fora_hip_get_batch(self._ctx)

    # ---- index
    def index_sizes(self):
        total = C.c_uint64(0)
        off = np.zeros(self.n, dtype=np.uint64)
        cnt = np.zeros(self.n, dtype=np.uint64)
        self._chk(self._lib.fora_hip_index_sizes(self._ctx, C.byref(total), _p(off), _p(cnt)))
        return total.value, off, cnt

    def build_index(self):
        self._chk(self._lib.fora_hip_build_index(self._ctx))

    def get_index(self):
        total, _, _ = self.index_sizes()
        rw = np.zeros(max(1, total), dtype=np.int32)
        off = np.zeros(self.n, dtype=np.uint64)
        cnt = np.zeros(self.n, dtype=np.uint64)
        self._chk(self._lib.fora_hip_get_index(self._ctx, _p(rw), C.c_uint64(total), _p(off), _p(cnt)))
        return rw[:total], off, cnt

    def set_index(self, rw_idx, off, cnt):
        rw_idx = np.ascontiguousarray(rw_idx, dtype=np.int32)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        cnt = np.ascontiguousarray(cnt, dtype=np.uint64)
        self._chk(self._lib.fora_hip_set_index(self._ctx, _p(rw_idx), C.c_uint64(rw_idx.size), _p(off), _p(cnt)))

    def clear_index(self):
        self._chk(self._lib.fora_hip_clear_index(self._ctx))

    # ---- queries
    @staticmethod
    def _stats(arr, n):
        # one structured-array copy instead of a dict per query (1000 queries: ~4 ms of getattr otherwise);
        # st[i]["pops"], iteration and len() work as they would on a list of dicts
        return np.frombuffer(arr, dtype=_STATS_DTYPE, count=n).copy()

    def query(self, sources, with_idx=False, want_ppr=True):
        src = np.ascontiguousarray(sources, dtype=np.int32)
        nq = src.size
        st = (QueryStats * max(1, nq))()
        out = np.zeros((nq, self.n), dtype=np.float64) if want_ppr else None
        self._chk(self._lib.fora_hip_query_batch(self._ctx, _p(src), C.c_int(nq), C.c_int(int(with_idx)),
                                                 _p(out), st))
        return out, self._stats(st, nq)

    def query_fix(self, sources, with_idx=False, want_residue=True):
        src = np.ascontiguousarray(sources, dtype=np.int32)
        nq = src.size
        st = (QueryStats * max(1, nq))()
        ppr = np.zeros((nq, self.n), dtype=np.uint64)
        res = np.zeros((nq, self.n), dtype=np.uint64) if want_residue else None
        self._chk(self._lib.fora_hip_query_batch_fix(self._ctx, _p(src), C.c_int(nq), C.c_int(int(with_idx)),
                                                     _p(ppr), _p(res), st))
        return ppr, res, self._stats(st, nq)

    def push(self, sources, want=True):
        """want=False: only the per-query stats come back (the slabs stay in HBM)."""
        src = np.ascontiguousarray(sources, dtype=np.int32)
        nq = src.size
        st = (QueryStats * max(1, nq))()
        rsv = np.zeros((nq, self.n), dtype=np.uint64) if want else None
        res = np.zeros((nq, self.n), dtype=np.uint64) if want else None
        self._chk(self._lib.fora_hip_push_batch(self._ctx, _p(src), C.c_int(nq), _p(rsv), _p(res), st))
        if not want:
            return self._stats(st, nq)
        return rsv, res, self._stats(st, nq)

    def topk(self, sources, k, epsilon=0.5, rmax_scale=1.0, with_idx=False):
        src = np.ascontiguousarray(sources, dtype=np.int32)
        nq = src.size
        ids = np.zeros((nq, k), dtype=np.int32)
        sc = np.zeros((nq, k), dtype=np.float64)
        rounds = np.zeros(max(1, nq), dtype=np.int32)
        self._chk(self._lib.fora_hip_topk_batch(self._ctx, _p(src), C.c_int(nq), C.c_int(k), C.c_double(epsilon),
                                                C.c_double(rmax_scale), C.c_int(int(with_idx)), _p(ids), _p(sc),
                                                _p(rounds)))
        return ids, sc, rounds[:nq]

    def set_balanced(self, on=True, start_scale=0.0, c_pop=0.0, c_edge=0.0, t_walk=0.0, t_idx=0.0):
        """--balanced (query.h:848-884); costs <= 0 select the MI355X defaults, start_scale <= 0 the reference's 8."""
        self._chk(self._lib.fora_hip_set_balanced(self._ctx, C.c_int(int(on)), C.c_double(start_scale), C.c_double(c_pop),
                                                  C.c_double(c_edge), C.c_double(t_walk), C.c_double(t_idx)))

    def topk_bound(self, sources, k, epsilon=0.5, rmax_scale=1.0, ppr_decay_alpha=0.77, with_idx=False):
        """get_topk without --opt (top-k with bounds, query.h:909-969)."""
        src = np.ascontiguousarray(sources, dtype=np.int32)
        nq = src.size
        ids = np.zeros((nq, k), dtype=np.int32)
        sc = np.zeros((nq, k), dtype=np.float64)
        rounds = np.zeros(max(1, nq), dtype=np.int32)
        self._chk(self._lib.fora_hip_topk_bound_batch(self._ctx, _p(src), C.c_int(nq), C.c_int(k), C.c_double(epsilon),
                                                      C.c_double(rmax_scale), C.c_double(ppr_decay_alpha),
                                                      C.c_int(int(with_idx)), _p(ids), _p(sc), _p(rounds)))
        return ids, sc, rounds[:nq]

    def power_iteration(self, sources, max_iter=100, k=0, want_ppr=True, want_fix=False):
        """Exact SSPPR (gen_exact_topk's fwd_power_iteration, query.h:1192-1238).  Returns (ppr f64 [nq,n] or
        None, ppr raw u64 or None, ids [nq,k] or None, scores or None)."""
        src = np.ascontiguousarray(sources, dtype=np.int32)
        nq = src.size
        ppr = np.zeros((nq, self.n), dtype=np.float64) if want_ppr else None
        fix = np.zeros((nq, self.n), dtype=np.uint64) if want_fix else None
        ids = np.zeros((nq, k), dtype=np.int32) if k else None
        sc = np.zeros((nq, k), dtype=np.float64) if k else None
        self._chk(self._lib.fora_hip_power_iteration_batch(
            self._ctx, _p(src), C.c_int(nq), C.c_int(max_iter), _p(ppr) if want_ppr else None,
            _p(fix) if want_fix else None, C.c_int(k), _p(ids) if k else None, _p(sc) if k else None))
        return ppr, fix, ids, sc

    # ---- stage hooks
    def walk_counts(self, residue, rsum):
        residue = np.ascontiguousarray(residue, dtype=np.float64)
        out = np.zeros(self.n, dtype=np.uint64)
        N = C.c_uint64(0)
        self._chk(self._lib.fora_hip_walk_counts(self._ctx, _p(residue), C.c_double(rsum), _p(out), C.byref(N)))
        return N.value, out

    def walks(self, stream, rnd, starts, js, no_zero_hop=False):
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        js = np.ascontiguousarray(js, dtype=np.uint64)
        out = np.zeros(starts.size, dtype=np.int32)
        self._chk(self._lib.fora_hip_walks(self._ctx, C.c_uint32(stream), C.c_uint32(rnd), C.c_int(int(no_zero_hop)),
                                           _p(starts), _p(js), C.c_int64(starts.size), _p(out)))
        return out

    # ---- measurement
    def set_option(self, name, value):
        self._chk(self._lib.fora_hip_set_option(self._ctx, name.encode(), C.c_int64(int(value))))

    def get_option(self, name):
        v = C.c_int64(0)
        self._chk(self._lib.fora_hip_get_option(self._ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def reset_options(self):
        self._chk(self._lib.fora_hip_set_option(self._ctx, b"reset", C.c_int64(0)))

    def reset_timing(self):
        self._chk(self._lib.fora_hip_reset_timing(self._ctx))

    def timing(self):
        t = Timing()
        self._chk(self._lib.fora_hip_get_timing(self._ctx, C.byref(t)))
        return t.as_dict()

    def stamps(self):
        """Diagnostic builds (-DFORA_STAMPS): cycles per kernel phase; zeros otherwise."""
        out = np.zeros(32, dtype=np.uint64)
        self._chk(self._lib.fora_hip_get_stamps(self._ctx, _p(out)))
        return out
