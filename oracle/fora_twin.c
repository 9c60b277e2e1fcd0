/*
 * fora_twin.c -- CPU twin of the HIP path's schedule and arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY (see fora_oracle.h).  PARITY UNPINNED against a built
 * reference (none can be built here); this file restates the SAME algorithm as
 * fora_oracle.c (citations below) with the two changes the GPU design makes:
 *
 *   1. level-synchronous (Jacobi) frontier order instead of the FIFO of
 *      algo.h:980-1017 -- every node at/over threshold is popped in the same
 *      level, their increments land together, nodes that CROSS their threshold
 *      during the level form the next frontier;
 *   2. residue / reserve / ppr are unsigned 2^-62 fixed point (1.0 == 2^62), so
 *      every accumulation is an exact integer add: the result does not depend on
 *      the order in which a GPU's atomics land, total mass is conserved exactly
 *      (sum reserve + sum residue == 2^62), and the HIP path can be compared
 *      BIT FOR BIT with this file.
 *
 * Both changes keep the push invariant of algo.h:954-1018
 *   pi(s,.) = reserve + sum_v residue[v] * pi(v,.)
 * and the exit condition residue[v] < rmax*outdeg(v); tests check both families
 * against the power iteration of query.h:1192-1224.
 */
#include "fora_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static inline uint64_t mulshift62(uint64_t r, uint64_t a) { return (uint64_t)(((u128)r * a) >> 62); }
static inline double fix2d(uint64_t x) { return ldexp((double)x, -62); }

/* alpha as 2^-62 fixed point: exact for the double alpha (0.2 -> 922337203685477632). */
uint64_t orc_twin_alpha_fix(double alpha) { return (uint64_t)ldexp(alpha, 62); }

/* threshold unit T1 = ceil(rmax * 2^62); node threshold = T1 * outdeg (saturating),
 * the integer form of "residue/outdeg >= rmax" (algo.h:1012). */
uint64_t orc_twin_rmax_fix(double rmax) {
    double t = ceil(ldexp(rmax, 62));
    if (t >= 9223372036854775808.0) return UINT64_MAX >> 1;
    if (t < 1.0) return 1;
    return (uint64_t)t;
}

static inline uint64_t node_thr(uint64_t t1, int64_t deg) {
    if (deg == 0) return 1; /* residue/0 = +inf >= rmax for any residue > 0 */
    u128 p = (u128)t1 * (uint64_t)deg;
    return (p >> 64) ? UINT64_MAX : (uint64_t)p;
}

void orc_fix_to_double(const uint64_t *in, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) out[i] = fix2d(in[i]);
}

/* Bounded deferral of the HIP push (fora_kernels.h k_accum, option "defer"; default 0 = off): a node that crosses its
 * threshold in a level but ends that level with less than 2^defer times the threshold waits ONE more level before it
 * is popped -- it collects what that level sends it, and is then popped whatever it holds.  A level-synchronous push
 * pops a node the level after it crosses; the FIFO of algo.h:980-1017 lets it collect more first and so moves more
 * mass per relaxed edge.  One level of patience for the nodes that only just crossed gives most of that back (ws-sized
 * bench graph: 1.34x -> 1.15x of the FIFO's relaxations, 1.10x -> 1.02x of its rsum, i.e. fewer walks), with no sweep
 * of the slab and without a second hump of large levels -- but with about twice the levels, which is what a level costs
 * on the GPU, hence off by default.  Exit condition and invariants are those of algo.h:1012.
 * Not used by the capped runs (power iteration) nor together with threshold rounds. */
static int g_twin_defer = 0;
void orc_twin_set_defer(int k) { g_twin_defer = k < 0 ? 0 : k > 8 ? 8 : k; }
int orc_twin_get_defer(void) { return g_twin_defer; }
static int64_t g_twin_defer_min = 0; /* only levels that pop at least this many nodes defer (option "defer_min") */
void orc_twin_set_defer_min(int64_t m) { g_twin_defer_min = m < 0 ? 0 : m; }
static int g_twin_rounds = 1;
static int g_twin_round_div = 0;

/* Runs levels until the frontier (and the set of deferred nodes) is empty.  frontier holds fn nodes on entry.  With switch_div > 0 (threshold
 * rounds) it also stops before a level whose frontier has shrunk to 1/switch_div of the largest one of this call;
 * the nodes of that frontier are still unpopped (they keep their residue).  Returns the size of the frontier left. */
static int64_t twin_levels_div(const int64_t *row_ptr, const int32_t *col, int32_t s, uint64_t t1,
                               uint64_t afix, uint64_t *residue, uint64_t *reserve, int32_t *frontier,
                               int64_t fn, int32_t *next, uint64_t *inc, orc_twin_push_stats *st,
                               int64_t *level_sizes, int64_t cap, int64_t max_levels, int64_t switch_div) {
    int64_t peak = 0;
    const int dk = (max_levels <= 0 && switch_div == 0 && g_twin_rounds == 1) ? g_twin_defer : 0;
    /* due: deferred by the previous level, they join the next frontier; wait: deferred by this level; cross: nodes
     * that crossed in this level, classified once the level's adds are complete (no array is longer than the number
     * of nodes: a node is in at most one of frontier / due / wait, and crosses at most once per level) */
    int32_t *due = NULL, *wait = NULL, *cross = NULL;
    int64_t ndue = 0, nalloc = 0;
    while ((fn > 0 || ndue > 0) && (max_levels <= 0 || st->levels < max_levels)) {
        if (switch_div > 0 && fn * switch_div <= peak) return fn;
        if (fn > peak) peak = fn;
        if (fn > 0) {
            if (level_sizes && st->levels < cap) level_sizes[st->levels] = fn;
            st->levels++; /* levels in which something was popped (the HIP path counts the same) */
        }
        uint64_t dang = 0;
        /* pop phase (algo.h:983-992, 1002): every frontier node gives up its residue */
        for (int64_t i = 0; i < fn; i++) {
            int32_t v = frontier[i];
            uint64_t r = residue[v];
            residue[v] = 0;
            uint64_t a = mulshift62(r, afix); /* v_residue * alpha */
            uint64_t push = r - a;            /* (1-alpha) * v_residue */
            int64_t deg = row_ptr[v + 1] - row_ptr[v];
            st->pops++;
            if (deg == 0) { /* algo.h:993-994: dangling mass goes to the source */
                reserve[v] += a;
                dang += push;
                inc[i] = 0;
            } else {
                uint64_t q = push / (uint64_t)deg;
                reserve[v] += a + (push - q * (uint64_t)deg); /* division remainder stays reserved */
                inc[i] = q;
            }
        }
        /* expand phase (algo.h:1003-1016) with threshold-crossing detection */
        int64_t nn = 0, ncross = 0;
        for (int64_t i = 0; i < fn; i++) {
            int32_t v = frontier[i];
            uint64_t q = inc[i];
            for (int64_t e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
                int32_t w = col[e];
                uint64_t old = residue[w], nw = old + q;
                residue[w] = nw;
                st->relax++;
                uint64_t thr = node_thr(t1, row_ptr[w + 1] - row_ptr[w]);
                if (old < thr && nw >= thr) {
                    if (!dk) next[nn++] = w;
                    else {
                        if (ncross == nalloc) { nalloc = nalloc ? nalloc * 2 : 1024; cross = (int32_t *)realloc(cross, sizeof(int32_t) * (size_t)nalloc); }
                        cross[ncross++] = w;
                    }
                }
            }
        }
        if (dang) { /* algo.h:994-998 */
            uint64_t old = residue[s], nw = old + dang;
            residue[s] = nw;
            uint64_t thr = node_thr(t1, row_ptr[s + 1] - row_ptr[s]);
            if (old < thr && nw >= thr) {
                if (!dk) next[nn++] = s;
                else {
                    if (ncross == nalloc) { nalloc = nalloc ? nalloc * 2 : 1024; cross = (int32_t *)realloc(cross, sizeof(int32_t) * (size_t)nalloc); }
                    cross[ncross++] = s;
                }
            }
        }
        if (dk) {
            /* the nodes deferred one level ago are due: next frontier, with whatever they hold when they are popped */
            for (int64_t i = 0; i < ndue; i++) next[nn++] = due[i];
            /* this level's crossing nodes: pop next level, or wait one level if they only just crossed */
            int64_t nwait = 0;
            if (ncross) wait = (int32_t *)realloc(wait, sizeof(int32_t) * (size_t)ncross);
            for (int64_t i = 0; i < ncross; i++) {
                const int32_t w = cross[i];
                const uint64_t thr = node_thr(t1, row_ptr[w + 1] - row_ptr[w]);
                if (fn >= g_twin_defer_min && (residue[w] >> dk) < thr) wait[nwait++] = w;
                else next[nn++] = w;
            }
            int32_t *t = due; due = wait; wait = t;
            ndue = nwait;
        }
        memcpy(frontier, next, sizeof(int32_t) * (size_t)nn);
        fn = nn;
    }
    free(due); free(wait); free(cross);
    return fn;
}

static void twin_levels(const int64_t *row_ptr, const int32_t *col, int32_t s, uint64_t t1,
                        uint64_t afix, uint64_t *residue, uint64_t *reserve, int32_t *frontier,
                        int64_t fn, int32_t *next, uint64_t *inc, orc_twin_push_stats *st,
                        int64_t *level_sizes, int64_t cap, int64_t max_levels) {
    (void)twin_levels_div(row_ptr, col, s, t1, afix, residue, reserve, frontier, fn, next, inc, st, level_sizes, cap,
                          max_levels, 0);
}

/* Threshold rounds of the HIP push (fora_kernels.h k_round_sweep): the levels first run against 2^(rounds-1) times the
 * threshold; whenever the frontier runs dry -- or, with round_div > 0, has shrunk to 1/round_div of the round's
 * largest frontier -- the threshold is halved and every node at or over the new one (found by a sweep over the
 * residues; this includes the nodes of the frontier that was left) forms the next frontier, down to the threshold of
 * algo.h:1012 itself.  rounds = 1 is the plain level-synchronous schedule (option "rounds" / "round_div"). */
void orc_twin_set_rounds(int rounds) { g_twin_rounds = rounds < 1 ? 1 : rounds > 16 ? 16 : rounds; }
int orc_twin_get_rounds(void) { return g_twin_rounds; }
void orc_twin_set_round_div(int div) { g_twin_round_div = div < 0 ? 0 : div; }
int orc_twin_get_round_div(void) { return g_twin_round_div; }
static uint64_t thr_unit(uint64_t t1, int k) { return k == 0 ? t1 : ((t1 >> (63 - k)) ? (UINT64_MAX >> 1) : (t1 << k)); }

/* Twin of forward_local_update_linear (algo.h:954-1018). */
int orc_twin_push(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                  double alpha, uint64_t *residue, uint64_t *ppr, orc_twin_push_stats *st,
                  int64_t *level_sizes, int64_t cap) {
    orc_twin_push_stats z = {0, 0, 0, 0};
    memset(residue, 0, sizeof(uint64_t) * (size_t)n);
    memset(ppr, 0, sizeof(uint64_t) * (size_t)n);
    if (row_ptr[s + 1] == row_ptr[s]) { /* algo.h:961-965 */
        ppr[s] = ORC_FIX_ONE;
        z.rsum_fix = 0;
        if (st) *st = z;
        return 0;
    }
    int32_t *frontier = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    uint64_t *inc = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    residue[s] = ORC_FIX_ONE; /* algo.h:976 */
    frontier[0] = s;          /* the source is pushed unconditionally (algo.h:973,980) */
    {
        const uint64_t t1 = orc_twin_rmax_fix(rmax), afix = orc_twin_alpha_fix(alpha);
        int64_t fn = 1;
        for (int k = g_twin_rounds - 1; k >= 0; k--) {
            const uint64_t unit = thr_unit(t1, k);
            if (k != g_twin_rounds - 1) { /* next round: every node at or over the halved threshold */
                fn = 0;
                for (int32_t v = 0; v < n; v++)
                    if (residue[v] && residue[v] >= node_thr(unit, row_ptr[v + 1] - row_ptr[v])) frontier[fn++] = v;
            }
            /* the last round runs dry; earlier ones may be left early (their pending nodes are found again by the sweep) */
            (void)twin_levels_div(row_ptr, col, s, unit, afix, residue, ppr, frontier, fn, next, inc, &z, level_sizes, cap, 0,
                                  k > 0 ? g_twin_round_div : 0);
        }
    }
    uint64_t reserved = 0;
    for (int32_t v = 0; v < n; v++) reserved += ppr[v];
    z.rsum_fix = ORC_FIX_ONE - reserved; /* == sum of residue, exactly */
    if (st) *st = z;
    free(frontier); free(next); free(inc);
    return 0;
}

/* Twin of fwd_power_iteration (query.h:1192-1224): max_iter levels of the same push with the smallest
 * threshold (one unit per out-edge) and no dangling-source short cut; ppr = what was reserved. */
int orc_twin_power_iteration(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double alpha,
                             int32_t max_iter, uint64_t *ppr, orc_twin_push_stats *st) {
    orc_twin_push_stats z = {0, 0, 0, 0};
    uint64_t *residue = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    int32_t *frontier = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    uint64_t *inc = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    memset(ppr, 0, sizeof(uint64_t) * (size_t)n);
    residue[s] = ORC_FIX_ONE;
    frontier[0] = s;
    twin_levels(row_ptr, col, s, 1, orc_twin_alpha_fix(alpha), residue, ppr, frontier, 1, next, inc, &z, NULL, 0,
                max_iter);
    uint64_t reserved = 0;
    for (int32_t v = 0; v < n; v++) reserved += ppr[v];
    z.rsum_fix = ORC_FIX_ONE - reserved;
    if (st) *st = z;
    free(residue); free(frontier); free(next); free(inc);
    return 0;
}

/* Twin of query.h:270,282,314 (and :349,:364 for --opt): per node walk counts. */
uint64_t orc_twin_walk_counts(int32_t n, const uint64_t *residue, uint64_t rsum_fix, double omega,
                              double alpha, int opt, uint64_t *num_s_rw) {
    memset(num_s_rw, 0, sizeof(uint64_t) * (size_t)n);
    if (rsum_fix == 0) return 0;
    double check_rsum = fix2d(rsum_fix);
    if (opt) check_rsum *= (1 - alpha);
    unsigned long long N = (unsigned long long)(omega * check_rsum);
    uint64_t afix = orc_twin_alpha_fix(alpha);
    for (int32_t v = 0; v < n; v++) {
        uint64_t r = residue[v];
        if (!r) continue;
        if (opt) r -= mulshift62(r, afix);
        num_s_rw[v] = (uint64_t)ceil(fix2d(r) / check_rsum * (double)N);
    }
    return N;
}

/* Twin of compute_ppr_with_fwdidx (query.h:255-327) / _opt (query.h:334-413).
 * ppr holds the reserve on entry and is refined in place.  Walk j of node v adds
 * r/num (+1 unit for the first r%num walks), so sum(ppr) stays exactly 2^62. */
int orc_twin_refine(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s,
                    const uint64_t *residue, uint64_t rsum_fix, double omega, double alpha, int opt,
                    uint64_t seed, const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                    uint64_t *ppr, orc_refine_stats *st) {
    orc_refine_stats z = {0, 0, 0};
    int64_t steps = 0;
    if (rsum_fix != 0) { /* query.h:267-268 */
        double check_rsum = fix2d(rsum_fix);
        if (opt) check_rsum *= (1 - alpha);                             /* query.h:349 */
        unsigned long long N = (unsigned long long)(omega * check_rsum); /* query.h:270 */
        uint64_t afix = orc_twin_alpha_fix(alpha);
        for (int32_t v = 0; v < n; v++) {
            uint64_t r = residue[v];
            if (!r) continue;
            if (opt) { /* query.h:363-364 */
                uint64_t a = mulshift62(r, afix);
                ppr[v] += a;
                r -= a;
            }
            uint64_t num = (uint64_t)ceil(fix2d(r) / check_rsum * (double)N); /* query.h:282 */
            if (num == 0) continue; /* reference: NaN weight, zero iterations */
            uint64_t incr = r / num, rem = r - incr * num;
            uint64_t from_idx = 0;
            if (rw_idx) { /* query.h:290-307 */
                from_idx = num > cnt[v] ? cnt[v] : num;
                for (uint64_t j = 0; j < from_idx; j++) ppr[rw_idx[off[v] + j]] += incr + (j < rem);
                z.n_idx_hit += from_idx;
            }
            for (uint64_t j = from_idx; j < num; j++) { /* query.h:320-323 */
                int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, 0, v, j, alpha, opt, &steps);
                ppr[des] += incr + (j < rem);
            }
            z.n_walks += num;
        }
    }
    z.walk_steps = (uint64_t)steps;
    if (st) *st = z;
    return 0;
}

/* Twin of fora_query_basic (query.h:841-907, non --balanced). */
int orc_twin_query(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                   double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                   const uint64_t *off, const uint64_t *cnt, uint64_t *residue, uint64_t *ppr,
                   orc_twin_push_stats *pst, orc_refine_stats *rst) {
    orc_twin_push_stats ps;
    orc_twin_push(n, row_ptr, col, s, rmax, alpha, residue, ppr, &ps, NULL, 0);
    orc_twin_refine(n, row_ptr, col, s, residue, ps.rsum_fix, omega, alpha, opt, seed, rw_idx, off,
                    cnt, ppr, rst);
    if (pst) *pst = ps;
    return 0;
}

typedef struct { int32_t id; uint64_t sc; } idfix;
static int cmp_idfix(const void *a, const void *b) {
    const idfix *x = (const idfix *)a, *y = (const idfix *)b;
    if (x->sc != y->sc) return (x->sc < y->sc) - (x->sc > y->sc);
    return (x->id > y->id) - (x->id < y->id);
}

/* Twin of fora_query_topk_new (query.h:972-1045) + topk_ppr (algo.h:592-610).
 * The incremental push (algo.h:1020-1093) restarts each round from every node whose
 * residue is at/over the round's threshold (the reference carries the same set in
 * forward_from whenever the round's rmax >= lowest_delta_rmax). */
int orc_twin_topk_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                        int32_t k, double epsilon, double alpha, double rmax_scale, uint64_t seed,
                        const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                        int32_t *ids, double *scores, int32_t *rounds, uint64_t *ppr_out) {
    if (k == 0) k = 500;
    const double min_delta = 1.0 / n;
    const double init_delta = 1.0 / k / 10;
    const double new_pfail = 1.0 / n / n;
    double pfail = new_pfail, delta = init_delta;
    uint64_t afix = orc_twin_alpha_fix(alpha);
    uint64_t *residue = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    uint64_t *reserve = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    uint64_t *ppr = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    uint64_t *cursor = rw_idx ? (uint64_t *)calloc((size_t)n, sizeof(uint64_t)) : NULL;
    int32_t *frontier = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    uint64_t *inc = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    int32_t nround = 0;
    residue[s] = ORC_FIX_ONE;

    while (delta >= min_delta) {
        double rmax, omega;
        orc_fora_topk_setting(m, epsilon, delta, pfail, rmax_scale, &rmax, &omega);
        nround++;
        if (row_ptr[s + 1] == row_ptr[s]) { /* query.h:1007-1011 */
            residue[s] = 0;
            reserve[s] = ORC_FIX_ONE;
            memcpy(ppr, reserve, sizeof(uint64_t) * (size_t)n);
            break;
        }
        uint64_t t1 = orc_twin_rmax_fix(rmax);
        int64_t fn = 0;
        for (int32_t v = 0; v < n; v++)
            if (residue[v] >= node_thr(t1, row_ptr[v + 1] - row_ptr[v])) frontier[fn++] = v;
        orc_twin_push_stats ps = {0, 0, 0, 0};
        twin_levels(row_ptr, col, s, t1, afix, residue, reserve, frontier, fn, next, inc, &ps, NULL, 0, 0);
        uint64_t reserved = 0;
        for (int32_t v = 0; v < n; v++) reserved += reserve[v];
        uint64_t rsum_fix = ORC_FIX_ONE - reserved;
        memcpy(ppr, reserve, sizeof(uint64_t) * (size_t)n); /* query.h:243-253 */
        if (rsum_fix != 0) {
            for (int32_t v = 0; v < n; v++) {
                uint64_t r = residue[v];
                if (!r) continue;
                if (rw_idx) { /* query.h:558-611 */
                    uint64_t a = mulshift62(r, afix);
                    ppr[v] += a;
                    r -= a;
                    uint64_t num = (uint64_t)ceil(fix2d(r) * omega);
                    if (!num) continue;
                    uint64_t incr = r / num, rem = r - incr * num;
                    uint64_t used = cursor[v], remaining = cnt[v] - used;
                    uint64_t from_idx = num <= remaining ? num : remaining;
                    for (uint64_t j = 0; j < from_idx; j++) ppr[rw_idx[off[v] + used + j]] += incr + (j < rem);
                    cursor[v] = used + from_idx;
                    for (uint64_t j = from_idx; j < num; j++) {
                        int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, (uint32_t)nround, v, j, alpha, 1, NULL);
                        ppr[des] += incr + (j < rem);
                    }
                } else { /* query.h:615-632 */
                    uint64_t num = (uint64_t)ceil(fix2d(r) * omega);
                    if (!num) continue;
                    uint64_t incr = r / num, rem = r - incr * num;
                    for (uint64_t j = 0; j < num; j++) {
                        int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, (uint32_t)nround, v, j, alpha, 0, NULL);
                        ppr[des] += incr + (j < rem);
                    }
                }
            }
        }
        /* algo.h:578-590 + query.h:1030: kth >= (1+eps)*delta  <=>  at least k entries >= it */
        double T = (1 + epsilon) * delta;
        int64_t above = 0;
        for (int32_t v = 0; v < n; v++) above += fix2d(ppr[v]) >= T;
        if (above >= k || delta <= min_delta) break;
        delta = delta / 4.0 > min_delta ? delta / 4.0 : min_delta;
    }
    idfix *all = (idfix *)malloc(sizeof(idfix) * ((size_t)n + 1));
    int64_t na = 0;
    for (int32_t v = 0; v < n; v++)
        if (ppr[v]) { all[na].id = v; all[na].sc = ppr[v]; na++; }
    qsort(all, (size_t)na, sizeof(idfix), cmp_idfix);
    for (int32_t i = 0; i < k; i++) {
        if (i < na) { ids[i] = all[i].id; scores[i] = fix2d(all[i].sc); }
        else { ids[i] = 0; scores[i] = 0.0; }
    }
    if (rounds) *rounds = nround;
    if (ppr_out) memcpy(ppr_out, ppr, sizeof(uint64_t) * (size_t)n);
    free(all); free(residue); free(reserve); free(ppr); free(cursor); free(frontier); free(next); free(inc);
    return 0;
}

/* ---- top-k with bounds (get_topk without --opt, query.h:909-969) in the twin's schedule ---------------------
 * Push / walks as in orc_twin_topk_query; the bounds (algo.h:1178-1261) and the stop rule (algo.h:1096-1166) are
 * f64 expressions of fix2d(ppr), fix2d(reserve) evaluated in the reference's operand order with L = log(2/pfail)
 * computed once -- +,*,/ and sqrt only per node, so the GPU reproduces them bit for bit. */
static inline double twin_lambda(double rsum, double L, double upper_bound, double total) {
    /* algo.h:1169-1174; total = (double)total_rw_num, 8*total exact below 2^50 */
    return 1.0 / 3 * L * rsum / total + sqrt(4.0 / 9.0 * L * L * rsum * rsum + 8 * total * L * rsum * upper_bound) / 2.0 / total;
}
void orc_twin_bounds_node(double p, double reserve, double rsum, double L, double total, double min_ppr,
                          double sqrt_min_ppr, double *upper, double *lower) {
    const double epsilon_v_div = sqrt(2.67 * rsum * L / total);
    const double default_epsilon_v = epsilon_v_div / sqrt_min_ppr;
    const double up0 = *upper, lo0 = *lower;
    double epsilon_a;
    if (up0 > reserve) epsilon_a = twin_lambda(rsum, L, up0 - reserve, total);
    else epsilon_a = twin_lambda(rsum, L, 1 - reserve, total);
    const double ub_eps_a = p + epsilon_a;
    double lb_eps_a = p - epsilon_a;
    if (!(lb_eps_a > 0)) lb_eps_a = 0;
    double epsilon_v = default_epsilon_v;
    if (reserve > 0 && reserve > min_ppr) {
        reserve = reserve > lo0 ? reserve : lo0;
        epsilon_v = epsilon_v_div / sqrt(reserve);
    } else if (lo0 > 0) {
        epsilon_v = epsilon_v_div / sqrt(lo0);
    }
    double ub_eps_v = 1.0, lb_eps_v = 0.0;
    if (1.0 - epsilon_v > 0) {
        ub_eps_v = p / (1.0 - epsilon_v);
        lb_eps_v = p / (1.0 + epsilon_v);
    }
    double up_bound = ub_eps_a < ub_eps_v ? ub_eps_a : ub_eps_v;
    if (!(up_bound < 1.0)) up_bound = 1.0;
    double low_bound = lb_eps_a > lb_eps_v ? lb_eps_a : lb_eps_v;
    if (!(low_bound > reserve)) low_bound = reserve;
    if (up_bound > 0) *upper = up_bound;
    if (low_bound >= 0) *lower = low_bound;
}

typedef struct { int32_t id; double sc; } idlbt;
static int cmp_idlbt(const void *a, const void *b) {
    const idlbt *x = (const idlbt *)a, *y = (const idlbt *)b;
    if (x->sc != y->sc) return (x->sc < y->sc) - (x->sc > y->sc);
    return (x->id > y->id) - (x->id < y->id);
}

int orc_twin_topk_bound_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                              int32_t k, double epsilon, double alpha, double rmax_scale, double ppr_decay_alpha,
                              uint64_t seed, const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                              int32_t *ids, double *scores, int32_t *rounds, uint64_t *ppr_out,
                              double *upper_out, double *lower_out) {
    const double min_delta = 1.0 / n;
    const double init_delta = 1.0 / 4;
    const double threshold = (1.0 - ppr_decay_alpha) / pow(500, ppr_decay_alpha) / pow(n, 1 - ppr_decay_alpha);
    const double pfail = 1.0 / n / n / log(n);
    const double L = log(2 / pfail);
    const double min_ppr = 1.0 / n, sqrt_min_ppr = sqrt(1.0 / n);
    double delta = init_delta;
    uint64_t afix = orc_twin_alpha_fix(alpha);
    uint64_t *residue = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    uint64_t *reserve = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    uint64_t *ppr = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    uint64_t *cursor = rw_idx ? (uint64_t *)calloc((size_t)n, sizeof(uint64_t)) : NULL;
    double *upper = (double *)malloc(sizeof(double) * (size_t)n), *lower = (double *)calloc((size_t)n, sizeof(double));
    for (int32_t v = 0; v < n; v++) upper[v] = 1.0;
    int32_t *frontier = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    uint64_t *inc = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    idlbt *tb = (idlbt *)malloc(sizeof(idlbt) * (size_t)n);
    unsigned char *filter = (unsigned char *)malloc((size_t)n);
    int32_t nround = 0;
    residue[s] = ORC_FIX_ONE;

    while (delta >= min_delta) {
        double rmax = epsilon * sqrt(delta / 3 / m / L); /* fora_setting with the round's delta, algo.h:455-463 */
        rmax *= rmax_scale;
        double omega = (2 + epsilon) * L / delta / epsilon / epsilon;
        nround++;
        if (row_ptr[s + 1] == row_ptr[s]) { /* query.h:951-955 */
            residue[s] = 0;
            reserve[s] = ORC_FIX_ONE;
            memcpy(ppr, reserve, sizeof(uint64_t) * (size_t)n);
            break;
        }
        uint64_t t1 = orc_twin_rmax_fix(rmax);
        int64_t fn = 0;
        for (int32_t v = 0; v < n; v++)
            if (residue[v] >= node_thr(t1, row_ptr[v + 1] - row_ptr[v])) frontier[fn++] = v;
        orc_twin_push_stats ps = {0, 0, 0, 0};
        twin_levels(row_ptr, col, s, t1, afix, residue, reserve, frontier, fn, next, inc, &ps, NULL, 0, 0);
        uint64_t reserved = 0;
        for (int32_t v = 0; v < n; v++) reserved += reserve[v];
        uint64_t rsum_fix = ORC_FIX_ONE - reserved;
        memcpy(ppr, reserve, sizeof(uint64_t) * (size_t)n);
        if (rsum_fix != 0) {
            double check_rsum = fix2d(rsum_fix);
            unsigned long long N = (unsigned long long)(omega * check_rsum); /* query.h:645 */
            uint64_t total = 0;
            for (int32_t v = 0; v < n; v++) {
                uint64_t r = residue[v];
                if (!r) continue;
                uint64_t num = rw_idx ? (uint64_t)ceil(fix2d(r) * omega)                      /* query.h:659 */
                                      : (uint64_t)ceil(fix2d(r) / check_rsum * (double)N);     /* query.h:727 */
                if (!num) continue;
                total += num;
                uint64_t incr = r / num, rem = r - incr * num;
                uint64_t from_idx = 0;
                if (rw_idx) {
                    uint64_t used = cursor[v], remaining = cnt[v] - used;
                    from_idx = num <= remaining ? num : remaining;
                    for (uint64_t j = 0; j < from_idx; j++) ppr[rw_idx[off[v] + used + j]] += incr + (j < rem);
                    cursor[v] = used + from_idx;
                }
                for (uint64_t j = from_idx; j < num; j++) {
                    int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, (uint32_t)nround, v, j, alpha, 0, NULL);
                    ppr[des] += incr + (j < rem);
                }
            }
            if (delta < threshold) /* query.h:745-746 */
                for (int32_t v = 0; v < n; v++)
                    if (ppr[v])
                        orc_twin_bounds_node(fix2d(ppr[v]), fix2d(reserve[v]), check_rsum, L, (double)total, min_ppr,
                                             sqrt_min_ppr, &upper[v], &lower[v]);
        }
        /* if_stop, algo.h:1096-1166 */
        int stop = 0;
        {
            int64_t above = 0;
            const double T = 2.0 * delta;
            for (int32_t v = 0; v < n; v++) above += fix2d(ppr[v]) >= T;
            if (above >= k) stop = 1;
            else if (!(delta >= threshold)) {
                for (int32_t v = 0; v < n; v++) { tb[v].id = v; tb[v].sc = lower[v]; }
                qsort(tb, (size_t)n, sizeof(idlbt), cmp_idlbt);
                memset(filter, 0, (size_t)n);
                int ok = 1;
                const double error = 1.0 + epsilon;
                for (int32_t i = 0; i < k; i++) {
                    /* zero lower bounds sort last among equals by id; the device select pads them: either way
                     * the ratio is +inf and the test fails */
                    filter[tb[i].id] = 1;
                    if (upper[tb[i].id] / lower[tb[i].id] > error) ok = 0;
                }
                const double low_bound_k = tb[k - 1].sc;
                if (ok && low_bound_k <= delta) ok = 0;
                if (ok)
                    for (int32_t v = 0; v < n; v++) {
                        if (filter[v] || !ppr[v]) continue;
                        if (upper[v] > low_bound_k * error && !(upper[v] > (1 + epsilon) / (1 - epsilon) * lower[v])) { ok = 0; break; }
                    }
                stop = ok;
            }
        }
        if (stop || delta <= min_delta) break;
        delta = delta / 2.0 > min_delta ? delta / 2.0 : min_delta;
    }
    idfix *all = (idfix *)malloc(sizeof(idfix) * ((size_t)n + 1));
    int64_t na = 0;
    for (int32_t v = 0; v < n; v++)
        if (ppr[v]) { all[na].id = v; all[na].sc = ppr[v]; na++; }
    qsort(all, (size_t)na, sizeof(idfix), cmp_idfix);
    for (int32_t i = 0; i < k; i++) {
        if (i < na) { ids[i] = all[i].id; scores[i] = fix2d(all[i].sc); }
        else { ids[i] = 0; scores[i] = 0.0; }
    }
    if (rounds) *rounds = nround;
    if (ppr_out) memcpy(ppr_out, ppr, sizeof(uint64_t) * (size_t)n);
    if (upper_out) memcpy(upper_out, upper, sizeof(double) * (size_t)n);
    if (lower_out) memcpy(lower_out, lower, sizeof(double) * (size_t)n);
    free(all); free(residue); free(reserve); free(ppr); free(cursor); free(frontier); free(next); free(inc);
    free(upper); free(lower); free(tb); free(filter);
    return 0;
}

/* ---- --balanced (fora_query_basic, query.h:848-884) in the twin's schedule --------------------------------------
 * The reference halves rmax from 8*config.rmax while the estimated walk cost (query.h:825-838) exceeds the
 * wall-clock time the push has taken so far.  Wall-clock is not reproducible; this build charges the push by its
 * work counters instead: used = pops*c_pop + relax*c_edge (seconds), and keeps the reference's walk estimate
 * omega*rsum*(1-alpha)*t_walk (t_idx once rmax < config.rmax with an index).  Rounds are incremental pushes from
 * every node at/over the round's threshold, as in the top-k driver.  Returns the number of rounds; *rmax_out =
 * the last rmax pushed with. */
int orc_twin_query_balanced(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax0,
                            double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                            const uint64_t *off, const uint64_t *cnt, double start_scale, double c_pop, double c_edge,
                            double t_walk, double t_idx, uint64_t *residue, uint64_t *ppr, orc_twin_push_stats *pst,
                            orc_refine_stats *rst, double *rmax_out) {
    orc_twin_push_stats z = {0, 0, 0, 0};
    memset(residue, 0, sizeof(uint64_t) * (size_t)n);
    memset(ppr, 0, sizeof(uint64_t) * (size_t)n);
    int rounds = 0;
    double rmax = rmax0 * (start_scale > 0 ? start_scale : 8); /* query.h:862 */
    if (row_ptr[s + 1] == row_ptr[s]) { /* :864, :882: plain push of a dangling source */
        ppr[s] = ORC_FIX_ONE;
        z.rsum_fix = 0;
        rmax = rmax0;
    } else {
        int32_t *frontier = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
        int32_t *next = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
        uint64_t *inc = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
        uint64_t afix = orc_twin_alpha_fix(alpha);
        residue[s] = ORC_FIX_ONE;
        uint64_t rsum_fix = ORC_FIX_ONE;
        double used = 0;
        for (;;) {
            const double t = (!rw_idx || rmax >= rmax0) ? t_walk : t_idx;              /* query.h:825-838 */
            const double est = omega * fix2d(rsum_fix) * (1 - alpha) * t;
            if (!(est > used)) break;                                                  /* :866 */
            uint64_t t1 = orc_twin_rmax_fix(rmax);
            int64_t fn = 0;
            for (int32_t v = 0; v < n; v++)
                if (residue[v] && residue[v] >= node_thr(t1, row_ptr[v + 1] - row_ptr[v])) frontier[fn++] = v;
            orc_twin_push_stats ps = {0, 0, 0, 0};
            twin_levels(row_ptr, col, s, t1, afix, residue, ppr, frontier, fn, next, inc, &ps, NULL, 0, 0);
            z.levels += ps.levels; z.pops += ps.pops; z.relax += ps.relax;
            used = (double)z.pops * c_pop + (double)z.relax * c_edge;                  /* :870-872, by counters */
            uint64_t reserved = 0;
            for (int32_t v = 0; v < n; v++) reserved += ppr[v];
            rsum_fix = ORC_FIX_ONE - reserved;
            rounds++;
            rmax /= 2;                                                                 /* :875 */
        }
        rmax *= 2;                                                                     /* :877 */
        z.rsum_fix = rsum_fix;
        free(frontier); free(next); free(inc);
    }
    orc_refine_stats rs = {0, 0, 0};
    orc_twin_refine(n, row_ptr, col, s, residue, z.rsum_fix, omega, alpha, opt, seed, rw_idx, off, cnt, ppr, &rs);
    if (pst) *pst = z;
    if (rst) *rst = rs;
    if (rmax_out) *rmax_out = rmax;
    return rounds;
}
