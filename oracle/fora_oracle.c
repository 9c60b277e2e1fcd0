/*
 * fora_oracle.c -- CPU restatement of the reference FORA hot path (f64, FIFO order).
 *
 * TEST INFRASTRUCTURE ONLY (see fora_oracle.h).  PARITY UNPINNED: the reference
 * cannot be built in this image (Boost absent) and ships no golden vectors for
 * this path; pinned pieces are listed in fora_oracle.h.
 *
 * Every function cites the /root/reference file:line it follows.  Operand order
 * and grouping of every floating-point expression follow the cited line; build
 * with -ffp-contract=off (the reference's x86-64 -O3 build has no FMA).
 */
#include "fora_oracle.h"
#include <math.h>
#include <stdio.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>

/* ------------------------------------------------------------------ loaders */

/* graph.h:48-64 init_nm: skip to '=', read n; skip to '=', read m. */
int orc_read_attribute(const char *path, int32_t *n, int64_t *m) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int c;
    long long v;
    while ((c = fgetc(f)) != EOF && c != '=') {}
    if (c == EOF || fscanf(f, "%lld", &v) != 1) { fclose(f); return -2; }
    *n = (int32_t)v;
    while ((c = fgetc(f)) != EOF && c != '=') {}
    if (c == EOF || fscanf(f, "%lld", &v) != 1) { fclose(f); return -2; }
    *m = (int64_t)v;
    fclose(f);
    return 0;
}

/* graph.h:152-154: while (fscanf(fin, "%d%d", &t1, &t2) != EOF) */
int64_t orc_count_edges(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int a, b;
    int64_t ne = 0;
    while (fscanf(f, "%d%d", &a, &b) == 2) ne++;
    fclose(f);
    return ne;
}

int orc_read_edges(const char *path, int32_t *src, int32_t *dst, int64_t cap, int64_t *ne) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int a, b;
    int64_t k = 0;
    while (fscanf(f, "%d%d", &a, &b) == 2) {
        if (k >= cap) { fclose(f); return -3; }
        src[k] = a;
        dst[k] = b;
        k++;
    }
    fclose(f);
    *ne = k;
    return 0;
}

/* graph.h:155-159: assert ids < n; skip t1==t2; g[t1].push_back(t2) in file order. */
int64_t orc_build_csr(int32_t n, const int32_t *src, const int32_t *dst, int64_t ne,
                      int64_t *row_ptr, int32_t *col) {
    memset(row_ptr, 0, sizeof(int64_t) * ((size_t)n + 1));
    for (int64_t e = 0; e < ne; e++) {
        if (src[e] >= n || dst[e] >= n || src[e] < 0 || dst[e] < 0) return -1;
        if (src[e] == dst[e]) continue;
        row_ptr[src[e] + 1]++;
    }
    for (int32_t v = 0; v < n; v++) row_ptr[v + 1] += row_ptr[v];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    memcpy(cur, row_ptr, sizeof(int64_t) * (size_t)n);
    for (int64_t e = 0; e < ne; e++) {
        if (src[e] == dst[e]) continue;
        col[cur[src[e]]++] = dst[e];
    }
    free(cur);
    return row_ptr[n];
}

/* algo.h:511-522 load_ss_query: while(queryfile >> v) push_back */
int64_t orc_read_queries(const char *path, int32_t *out, int64_t cap) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int v;
    int64_t k = 0;
    while (k < cap && fscanf(f, "%d", &v) == 1) out[k++] = v;
    fclose(f);
    return k;
}

/* --------------------------------------------------------------- parameters */

/* graph.h:177-178 (delta = pfail = 1.0/n) + algo.h:455-463 fora_setting. */
void orc_fora_setting(int32_t n, int64_t m, double epsilon, double alpha, double rmax_scale,
                      int opt, double *rmax, double *omega) {
    double delta = 1.0 / n;
    double pfail = 1.0 / n;
    double r = epsilon * sqrt(delta / 3 / m / log(2 / pfail));
    if (opt)
        r *= rmax_scale / (1 - alpha);
    else
        r *= rmax_scale;
    *rmax = r;
    *omega = (2 + epsilon) * log(2 / pfail) / delta / epsilon / epsilon;
}

/* algo.h:466-474 fora_topk_setting (both branches are the same expression). */
void orc_fora_topk_setting(int64_t m, double epsilon, double delta, double pfail,
                           double rmax_scale, double *rmax, double *omega) {
    double r = epsilon * sqrt(delta / 3 / m / log(2 / pfail));
    r *= sqrt(1.0 * m * r) * rmax_scale * 3;
    *rmax = r;
    *omega = (2 + epsilon) * log(2 / pfail) / delta / epsilon / epsilon;
}

/* --------------------------------------------------------------- Philox RNG */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 constants).  Replaces the
 * time(0)-seeded Boost engines of algo.h:105-122, which are not reproducible. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Walk contract.  Semantics follow algo.h:124-142 (random_walk) and algo.h:144-166
 * (random_walk_no_zero_hop): a walk from a dangling start returns at once; the
 * Bernoulli(alpha) stop test comes BEFORE each move; a dangling current node
 * jumps back to the start and the walk goes on.
 * Randomness: one Philox call per two steps, counter =
 *   (start, j lo32, j bits 32..47 | round<<16 | (step/2 & 255)<<24, stream ^ (step>>9)*0x9E3779B9),
 * (the last word moves on every 512 steps: the 8-bit step field alone would repeat its draws with period 512,
 *  and a walk that survived one period would never stop -- certain at small alpha and 1e8+ walks),
 * key = (seed lo32, seed hi32).  Step t uses words 2(t&1) [stop if < floor(alpha*2^32)]
 * and 2(t&1)+1 [neighbour = (word*deg)>>32]. */
int32_t orc_walk(int32_t n, const int64_t *row_ptr, const int32_t *col, uint64_t seed,
                 uint32_t stream, uint32_t round, int32_t start, uint64_t j, double alpha,
                 int no_zero_hop, int64_t *steps) {
    (void)n;
    int64_t d0 = row_ptr[start + 1] - row_ptr[start];
    if (d0 == 0) return start;
    uint32_t alpha32 = (uint32_t)(alpha * 4294967296.0);
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    int32_t cur = start;
    for (uint32_t t = 0;; t++) {
        if ((t & 1u) == 0) {
            uint32_t ctr[4] = {(uint32_t)start, (uint32_t)j,
                               (uint32_t)((j >> 32) & 0xFFFFu) | ((round & 0xFFu) << 16) |
                                   (((t >> 1) & 0xFFu) << 24),
                               stream ^ ((t >> 9) * 0x9E3779B9u)};
            orc_philox4x32_10(ctr, key, w);
        }
        uint32_t ws = w[(t & 1u) * 2], wm = w[(t & 1u) * 2 + 1];
        if (!(no_zero_hop && t == 0) && ws < alpha32) return cur;
        int64_t b = row_ptr[cur], d = row_ptr[cur + 1] - b;
        if (d > 0)
            cur = col[b + (int64_t)(((uint64_t)wm * (uint64_t)d) >> 32)];
        else
            cur = start;
        if (steps) (*steps)++;
    }
}

/* ------------------------------------------------------------ FIFO push f64 */
/* algo.h:954-1018 forward_local_update_linear. */
/* scratch of one push: `idx` (n bytes) is all zero between calls (every node that enters the queue leaves it), `q` grows */
typedef struct { unsigned char *idx; int32_t *q; int64_t cap; } push_ws;
static int push_fifo_ws(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                        double alpha, double *reserve, double *residue, int32_t *reserve_occur,
                        int32_t *residue_occur, orc_push_stats *st, push_ws *ws);
int orc_push_fifo(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                  double alpha, double *reserve, double *residue, int32_t *reserve_occur,
                  int32_t *residue_occur, orc_push_stats *st) {
    push_ws ws;
    ws.idx = (unsigned char *)calloc((size_t)n, 1); /* :958-959 */
    ws.cap = (int64_t)n + 16;
    ws.q = (int32_t *)malloc(sizeof(int32_t) * (size_t)ws.cap);
    const int rc = push_fifo_ws(n, row_ptr, col, s, rmax, alpha, reserve, residue, reserve_occur, residue_occur, st, &ws);
    free(ws.idx); free(ws.q);
    return rc;
}
static int push_fifo_ws(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                        double alpha, double *reserve, double *residue, int32_t *reserve_occur,
                        int32_t *residue_occur, orc_push_stats *st, push_ws *ws) {
    const double nil = -1.0; /* query.h:1464-1465 */
    for (int32_t i = 0; i < n; i++) { reserve[i] = nil; residue[i] = nil; } /* :955-956 clean() */
    unsigned char *idx = ws->idx;                                            /* :958-959 */
    int64_t n_res = 0, n_rsd = 0, pops = 0, relax = 0, gens = 0;
    double rsum = 1.0; /* query.h:843 */

    if (row_ptr[s + 1] == row_ptr[s]) { /* :961-965 */
        reserve[s] = 1;
        reserve_occur[n_res++] = s;
        rsum = 0;
        goto done;
    }
    {
        const double myeps = rmax;
        int64_t cap = ws->cap, qn = 0, left = 0;
        int32_t *q = ws->q;
        q[qn++] = s;                               /* :969-973 (sentinel slot omitted) */
        residue[s] = 1.0; residue_occur[n_rsd++] = s; /* :976 insert(s, init_residual) */
        idx[s] = 1;
        int64_t gen_end = qn;
        while (left < qn) { /* :980 */
            if (left == gen_end) { gens++; gen_end = qn; }
            int32_t v = q[left];
            idx[v] = 0;
            left++;
            pops++;
            double v_residue = residue[v];
            residue[v] = 0;
            if (reserve[v] == nil) { /* :986-989 */
                reserve[v] = v_residue * alpha;
                reserve_occur[n_res++] = v;
            } else
                reserve[v] += v_residue * alpha;
            int64_t out_neighbor = row_ptr[v + 1] - row_ptr[v];
            rsum -= v_residue * alpha; /* :992 */
            if (out_neighbor == 0) {   /* :993-1000 */
                residue[s] += v_residue * (1 - alpha);
                int64_t ds = row_ptr[s + 1] - row_ptr[s];
                if (ds > 0 && residue[s] / (double)ds >= myeps && !idx[s]) {
                    idx[s] = 1;
                    if (qn == cap) { cap *= 2; q = (int32_t *)realloc(q, sizeof(int32_t) * (size_t)cap); }
                    q[qn++] = s;
                }
                continue;
            }
            double avg_push_residual = ((1.0 - alpha) * v_residue) / out_neighbor; /* :1002 */
            for (int64_t e = row_ptr[v]; e < row_ptr[v + 1]; e++) {                /* :1003 */
                int32_t next = col[e];
                relax++;
                if (residue[next] == nil) { /* :1005-1008 */
                    residue[next] = avg_push_residual;
                    residue_occur[n_rsd++] = next;
                } else
                    residue[next] += avg_push_residual;
                double dn = (double)(row_ptr[next + 1] - row_ptr[next]);
                if (residue[next] / dn >= myeps && !idx[next]) { /* :1012 */
                    idx[next] = 1;
                    if (qn == cap) { cap *= 2; q = (int32_t *)realloc(q, sizeof(int32_t) * (size_t)cap); }
                    q[qn++] = next;
                }
            }
        }
        gens++;
        ws->q = q; ws->cap = cap;
    }
done:
    if (st) {
        st->rsum = rsum; st->pops = pops; st->relax = relax;
        st->n_reserve = n_res; st->n_residue = n_rsd; st->generations = gens;
    }
    return 0;
}

/* ---------------------------------------------------------- walk allocation */
/* query.h:270 (N), :282/:314 (num_s_rw); --opt: query.h:349 (rsum*=1-alpha), :364 (r*(1-alpha)). */
uint64_t orc_walk_counts(const double *residue, const int32_t *residue_occur, int64_t n_residue,
                         double rsum, double omega, double alpha, int opt, uint64_t *num_s_rw) {
    double check_rsum = rsum;
    if (check_rsum == 0.0) {
        for (int64_t i = 0; i < n_residue; i++) num_s_rw[i] = 0;
        return 0;
    }
    if (opt) check_rsum *= (1 - alpha);
    unsigned long long num_random_walk = (unsigned long long)(omega * check_rsum);
    for (int64_t i = 0; i < n_residue; i++) {
        double residual = residue[residue_occur[i]];
        if (opt) residual = residual * (1 - alpha);
        num_s_rw[i] = (unsigned long)ceil(residual / check_rsum * num_random_walk);
    }
    return num_random_walk;
}

/* --------------------------------------------------------------- walk index */
/* build.h:325-334: num_rw = ceil(outdeg*rmax*omega) (x(1-alpha) between rmax and omega if --opt). */
uint64_t orc_index_sizes(int32_t n, const int64_t *row_ptr, double rmax, double omega, double alpha,
                         int opt, uint64_t *off, uint64_t *cnt) {
    uint64_t total = 0;
    for (int32_t v = 0; v < n; v++) {
        size_t deg = (size_t)(row_ptr[v + 1] - row_ptr[v]);
        unsigned long num_rw;
        if (opt)
            num_rw = (unsigned long)ceil(deg * rmax * (1 - alpha) * omega);
        else
            num_rw = (unsigned long)ceil(deg * rmax * omega);
        off[v] = total;
        cnt[v] = num_rw;
        total += num_rw;
    }
    return total;
}

/* build.h:344-354: for each source, cnt walks (random_walk_no_zero_hop if --opt). */
void orc_build_index(int32_t n, const int64_t *row_ptr, const int32_t *col, uint64_t seed,
                     double alpha, int opt, const uint64_t *off, const uint64_t *cnt,
                     int32_t *rw_idx) {
    for (int32_t v = 0; v < n; v++)
        for (uint64_t i = 0; i < cnt[v]; i++)
            rw_idx[off[v] + i] =
                orc_walk(n, row_ptr, col, seed, ORC_STREAM_INDEX, 0, v, i, alpha, opt, NULL);
}

/* --------------------------------------------------------------- refinement */
static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* query.h:255-327 compute_ppr_with_fwdidx and query.h:334-413 compute_ppr_with_fwdidx_opt. */
int orc_refine(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s,
               const double *reserve, const int32_t *reserve_occur, int64_t n_reserve,
               const double *residue, int32_t *residue_occur, int64_t n_residue, double rsum,
               double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
               const uint64_t *off, const uint64_t *cnt, double *ppr, orc_refine_stats *st) {
    uint64_t n_walks = 0, n_hit = 0;
    int64_t steps = 0;
    memset(ppr, 0, sizeof(double) * (size_t)n);                                      /* :256 */
    for (int64_t i = 0; i < n_reserve; i++) ppr[reserve_occur[i]] = reserve[reserve_occur[i]]; /* :260-264 */
    double check_rsum = rsum;
    if (check_rsum == 0.0) goto done; /* :267-268 */
    if (opt) check_rsum *= (1 - alpha); /* :349 */
    {
        unsigned long long num_random_walk = (unsigned long long)(omega * check_rsum); /* :270 */
        if (rw_idx) qsort(residue_occur, (size_t)n_residue, sizeof(int32_t), cmp_i32);  /* :278 Sort() */
        for (int64_t i = 0; i < n_residue; i++) {
            int32_t source = residue_occur[i];
            double residual = residue[source];
            if (opt) { /* :363-364 */
                ppr[source] += residue[source] * alpha;
                residual = residue[source] * (1 - alpha);
            }
            unsigned long num_s_rw = (unsigned long)ceil(residual / check_rsum * num_random_walk); /* :282 */
            double a_s = residual / check_rsum * num_random_walk / num_s_rw;                      /* :283 */
            double ppr_incre = a_s * check_rsum / num_random_walk;                                /* :285 */
            n_walks += num_s_rw;                                                                  /* :287 */
            uint64_t from_idx = 0;
            if (rw_idx) { /* :290-307 */
                from_idx = num_s_rw > cnt[source] ? cnt[source] : num_s_rw;
                for (uint64_t k = 0; k < from_idx; k++) ppr[rw_idx[off[source] + k]] += ppr_incre;
                n_hit += from_idx;
            }
            for (uint64_t j = from_idx; j < num_s_rw; j++) { /* :297-300 / :320-323 */
                int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, 0, source, j, alpha, opt, &steps);
                ppr[des] += ppr_incre;
            }
        }
    }
done:
    if (st) { st->n_walks = n_walks; st->n_idx_hit = n_hit; st->walk_steps = (uint64_t)steps; }
    return 0;
}

/* query.h:841-907 fora_query_basic, non --balanced branch (:886, :894-900). */
int orc_query(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
              double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
              const uint64_t *off, const uint64_t *cnt, double *ppr, orc_push_stats *pst,
              orc_refine_stats *rst) {
    double *reserve = (double *)malloc(sizeof(double) * (size_t)n);
    double *residue = (double *)malloc(sizeof(double) * (size_t)n);
    int32_t *o1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *o2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    orc_push_stats ps;
    orc_push_fifo(n, row_ptr, col, s, rmax, alpha, reserve, residue, o1, o2, &ps);
    orc_refine(n, row_ptr, col, s, reserve, o1, ps.n_reserve, residue, o2, ps.n_residue, ps.rsum,
               omega, alpha, opt, seed, rw_idx, off, cnt, ppr, rst);
    if (pst) *pst = ps;
    free(reserve); free(residue); free(o1); free(o2);
    return 0;
}

/* The query() loop of query.h:1471-1476 over T host threads for bench.py's all-core baseline: thread t takes sources
 * t, t + T, ... until `seconds` have passed; every thread owns its buffers for the whole run (per-call malloc / free of
 * n-sized arrays is an mmap / munmap pair each, which serialised 256 Python threads on the kernel's address-space lock:
 * 10x one thread).  Returns the queries finished; *elapsed = wall time of the slowest thread. */
typedef struct {
    int32_t n; const int64_t *row_ptr; const int32_t *col; const int32_t *sources; int64_t nsrc;
    double rmax, omega, alpha; int opt; uint64_t seed;
    const int32_t *rw_idx; const uint64_t *off, *cnt;
    int tid, threads; double seconds;
    int cpu; /* >= 0: the worker binds itself to this host CPU before it allocates its buffers */
    int64_t done; uint64_t walks; double elapsed;
} many_arg;
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void *many_worker(void *p) {
    many_arg *a = (many_arg *)p;
    const size_t n = (size_t)a->n;
    if (a->cpu >= 0) { /* one thread per allowed CPU, first touch of its buffers on that CPU's memory node */
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(a->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    double *reserve = (double *)malloc(sizeof(double) * n), *residue = (double *)malloc(sizeof(double) * n);
    double *ppr = (double *)malloc(sizeof(double) * n);
    int32_t *o1 = (int32_t *)malloc(sizeof(int32_t) * n), *o2 = (int32_t *)malloc(sizeof(int32_t) * n);
    push_ws ws;
    ws.idx = (unsigned char *)calloc(n, 1);
    ws.cap = (int64_t)n + 16;
    ws.q = (int32_t *)malloc(sizeof(int32_t) * (size_t)ws.cap);
    const double t0 = now_s();
    for (int64_t i = a->tid; i < a->nsrc; i += a->threads) {
        orc_push_stats ps;
        orc_refine_stats rs;
        push_fifo_ws(a->n, a->row_ptr, a->col, a->sources[i], a->rmax, a->alpha, reserve, residue, o1, o2, &ps, &ws);
        orc_refine(a->n, a->row_ptr, a->col, a->sources[i], reserve, o1, ps.n_reserve, residue, o2, ps.n_residue, ps.rsum,
                   a->omega, a->alpha, a->opt, a->seed, a->rw_idx, a->off, a->cnt, ppr, &rs);
        a->done++;
        a->walks += rs.n_walks;
        if (now_s() - t0 > a->seconds) break;
    }
    a->elapsed = now_s() - t0;
    free(reserve); free(residue); free(ppr); free(o1); free(o2); free(ws.idx); free(ws.q);
    return NULL;
}
int64_t orc_query_many(int32_t n, const int64_t *row_ptr, const int32_t *col, const int32_t *sources, int64_t nsrc,
                       double rmax, double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                       const uint64_t *off, const uint64_t *cnt, int threads, double seconds, double *elapsed,
                       uint64_t *walks) {
    return orc_query_many_pinned(n, row_ptr, col, sources, nsrc, rmax, omega, alpha, opt, seed, rw_idx, off, cnt, threads, seconds,
                                 elapsed, walks, NULL, 0);
}
/* cpus[0 .. ncpus): host CPUs the threads bind to, thread t -> cpus[t mod ncpus] (NULL: the scheduler places them) */
int64_t orc_query_many_pinned(int32_t n, const int64_t *row_ptr, const int32_t *col, const int32_t *sources, int64_t nsrc,
                              double rmax, double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                              const uint64_t *off, const uint64_t *cnt, int threads, double seconds, double *elapsed,
                              uint64_t *walks, const int32_t *cpus, int ncpus) {
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    many_arg *args = (many_arg *)calloc((size_t)threads, sizeof(many_arg));
    for (int t = 0; t < threads; t++) {
        many_arg *a = &args[t];
        a->n = n; a->row_ptr = row_ptr; a->col = col; a->sources = sources; a->nsrc = nsrc;
        a->rmax = rmax; a->omega = omega; a->alpha = alpha; a->opt = opt; a->seed = seed;
        a->rw_idx = rw_idx; a->off = off; a->cnt = cnt; a->tid = t; a->threads = threads; a->seconds = seconds;
        a->cpu = (cpus && ncpus > 0) ? cpus[t % ncpus] : -1;
        if (pthread_create(&th[t], NULL, many_worker, a) != 0) { threads = t; break; }
    }
    int64_t done = 0;
    uint64_t w = 0;
    double el = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        done += args[t].done; w += args[t].walks;
        if (args[t].elapsed > el) el = args[t].elapsed;
    }
    if (elapsed) *elapsed = el;
    if (walks) *walks = w;
    free(th); free(args);
    return done;
}

/* ------------------------------------------------------------------- top-k */
typedef struct {
    double *val;
    unsigned char *has;
    int32_t *occur;
    int64_t n_occur;
} smap; /* iMap<double>: dense array + existence + first-touch list, mylib.h:278-425 */

static void smap_init(smap *m, int32_t n) {
    m->val = (double *)calloc((size_t)n, sizeof(double));
    m->has = (unsigned char *)calloc((size_t)n, 1);
    m->occur = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    m->n_occur = 0;
}
static void smap_free(smap *m) { free(m->val); free(m->has); free(m->occur); }
static void smap_clean(smap *m) { /* mylib.h:315-323 */
    for (int64_t i = 0; i < m->n_occur; i++) { m->has[m->occur[i]] = 0; m->val[m->occur[i]] = 0; }
    m->n_occur = 0;
}
static void smap_insert(smap *m, int32_t p, double d) { /* mylib.h:387-399 */
    if (!m->has[p]) { m->has[p] = 1; m->occur[m->n_occur++] = p; }
    m->val[p] = d;
}
static void smap_add(smap *m, int32_t p, double d) { /* "if(!exist) insert else +=" idiom */
    if (!m->has[p]) smap_insert(m, p, d); else m->val[p] += d;
}

typedef struct { int32_t *a; int64_t n, cap; } ivec;
static void ivec_push(ivec *v, int32_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->a = (int32_t *)realloc(v->a, sizeof(int32_t) * (size_t)v->cap); }
    v->a[v->n++] = x;
}

/* algo.h:1020-1093 forward_local_update_linear_topk. */
static int64_t topk_push_pops, topk_push_relax; /* work counters of push_fifo_topk (bench.py's roofline of the top-k push; single-threaded use) */
static void push_fifo_topk(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s,
                           double *rsum, double rmax, double lowest_rmax, double alpha, smap *reserve,
                           smap *residue, ivec *forward_from, unsigned char *in_forward,
                           unsigned char *in_next_forward) {
    const double myeps = rmax;
    memset(in_forward, 0, (size_t)n);      /* :1026 */
    memset(in_next_forward, 0, (size_t)n); /* :1027 */
    ivec next = {0, 0, 0};
    for (int64_t i = 0; i < forward_from->n; i++) in_forward[forward_from->a[i]] = 1; /* :1031-1032 */
    int64_t i = 0;
    while (i < forward_from->n) { /* :1035 */
        int32_t v = forward_from->a[i];
        i++;
        in_forward[v] = 0;
        double dv = (double)(row_ptr[v + 1] - row_ptr[v]);
        if (residue->val[v] / dv >= myeps) { /* :1039 */
            int64_t out_neighbor = row_ptr[v + 1] - row_ptr[v];
            double v_residue = residue->val[v];
            residue->val[v] = 0;
            topk_push_pops++; topk_push_relax += out_neighbor;
            smap_add(reserve, v, v_residue * alpha); /* :1043-1048 */
            *rsum -= v_residue * alpha;              /* :1050 */
            if (out_neighbor == 0) {                 /* :1051-1064 */
                residue->val[s] += v_residue * (1 - alpha);
                double ds = (double)(row_ptr[s + 1] - row_ptr[s]);
                if (ds > 0 && !in_forward[s] && residue->val[s] / ds >= myeps) {
                    ivec_push(forward_from, s);
                    in_forward[s] = 1;
                } else if (!in_next_forward[s] && residue->val[s] / ds >= lowest_rmax) {
                    ivec_push(&next, s);
                    in_next_forward[s] = 1;
                }
                continue;
            }
            double avg_push_residual = ((1 - alpha) * v_residue) / out_neighbor; /* :1066 */
            for (int64_t e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
                int32_t nx = col[e];
                smap_add(residue, nx, avg_push_residual); /* :1068-1071 */
                double dn = (double)(row_ptr[nx + 1] - row_ptr[nx]);
                if (!in_forward[nx] && residue->val[nx] / dn >= myeps) { /* :1073 */
                    ivec_push(forward_from, nx);
                    in_forward[nx] = 1;
                } else if (!in_next_forward[nx] && residue->val[nx] / dn >= lowest_rmax) { /* :1078 */
                    ivec_push(&next, nx);
                    in_next_forward[nx] = 1;
                }
            }
        } else if (!in_next_forward[v] && residue->val[v] / dv >= lowest_rmax) { /* :1085-1088 */
            ivec_push(&next, v);
            in_next_forward[v] = 1;
        }
    }
    free(forward_from->a); /* :1092 forward_from = next_forward_from */
    *forward_from = next;
}

static int cmp_desc_d(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x < y) - (x > y);
}
typedef struct { int32_t id; double sc; } idsc;
static int cmp_idsc(const void *a, const void *b) {
    const idsc *x = (const idsc *)a, *y = (const idsc *)b;
    if (x->sc != y->sc) return (x->sc < y->sc) - (x->sc > y->sc);
    return (x->id > y->id) - (x->id < y->id); /* tie order is unspecified in algo.h:605; id asc here */
}

/* query.h:972-1045 fora_query_topk_new + algo.h:592-610 topk_ppr. */
int orc_topk_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                   int32_t k, double epsilon, double alpha, double rmax_scale, uint64_t seed,
                   const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                   int32_t *ids, double *scores, int32_t *rounds, double *ppr_out) {
    if (k == 0) k = 500;                          /* :975 */
    const double min_delta = 1.0 / n;             /* :974 */
    const double init_delta = 1.0 / k / 10;       /* :976 */
    const double new_pfail = 1.0 / n / n;         /* :977 */
    double pfail = new_pfail, delta = init_delta; /* :979-980 */
    const double lowest_delta_rmax = epsilon * sqrt(min_delta / 3 / m / log(2 / new_pfail)); /* :982 */
    double rsum = 1.0;
    smap reserve, residue, ppr;
    smap_init(&reserve, n); smap_init(&residue, n); smap_init(&ppr, n);
    ivec forward_from = {0, 0, 0};
    ivec_push(&forward_from, s);    /* :989 */
    smap_insert(&residue, s, rsum); /* :993 */
    uint64_t *rw_counter = rw_idx ? (uint64_t *)calloc((size_t)n, sizeof(uint64_t)) : NULL; /* :997-998 */
    unsigned char *f1 = (unsigned char *)malloc((size_t)n), *f2 = (unsigned char *)malloc((size_t)n);
    int32_t nround = 0;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);

    while (delta >= min_delta) { /* :1001 */
        double rmax, omega;
        orc_fora_topk_setting(m, epsilon, delta, pfail, rmax_scale, &rmax, &omega); /* :1002 */
        nround++;
        if (row_ptr[s + 1] == row_ptr[s]) { /* :1007-1011 */
            rsum = 0.0;
            smap_insert(&reserve, s, 1);
            smap_clean(&ppr);
            smap_insert(&ppr, s, 1); /* compute_ppr_with_reserve query.h:243-253 */
            break;
        }
        push_fifo_topk(n, row_ptr, col, s, &rsum, rmax, lowest_delta_rmax, alpha, &reserve, &residue,
                       &forward_from, f1, f2); /* :1013 */

        /* query.h:521-636 compute_ppr_with_fwdidx_topk */
        smap_clean(&ppr); /* :532 -> :243-253 */
        for (int64_t i = 0; i < reserve.n_occur; i++) {
            int32_t v = reserve.occur[i];
            if (reserve.val[v]) smap_insert(&ppr, v, reserve.val[v]);
        }
        if (rsum != 0.0) { /* :535-536 */
            if (rw_idx) {  /* :555-613 */
                qsort(residue.occur, (size_t)residue.n_occur, sizeof(int32_t), cmp_i32); /* :556 */
                for (int64_t i = 0; i < residue.n_occur; i++) {
                    int32_t source = residue.occur[i];
                    double residual = residue.val[source];
                    smap_add(&ppr, source, residual * alpha);               /* :561-565 */
                    residual *= (1 - alpha);                                /* :567 */
                    unsigned long num_s_rw = (unsigned long)ceil(residual * omega); /* :568 */
                    double a_s = residual * omega / num_s_rw;               /* :569 */
                    double ppr_incre = a_s / omega;                         /* :571 */
                    uint64_t used = rw_counter[source];                     /* :575 */
                    uint64_t remaining = cnt[source] - used;                /* :576 */
                    uint64_t from_idx = num_s_rw <= remaining ? num_s_rw : remaining;
                    for (uint64_t kk = 0; kk < from_idx; kk++)              /* :580-586 / :594-600 */
                        smap_add(&ppr, rw_idx[off[source] + used + kk], ppr_incre);
                    rw_counter[source] = used + from_idx;                   /* :588 / :603 */
                    for (uint64_t j = from_idx; j < num_s_rw; j++) {        /* :605-611 */
                        int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, (uint32_t)nround,
                                               source, j, alpha, 1, NULL);
                        smap_add(&ppr, des, ppr_incre);
                    }
                }
            } else { /* :615-632 */
                for (int64_t i = 0; i < residue.n_occur; i++) {
                    int32_t source = residue.occur[i];
                    double residual = residue.val[source];
                    unsigned long num_s_rw = (unsigned long)ceil(residual * omega); /* :618 */
                    double a_s = residual * omega / num_s_rw;
                    double ppr_incre = a_s / omega;
                    for (uint64_t j = 0; j < num_s_rw; j++) {
                        int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, (uint32_t)nround,
                                               source, j, alpha, 0, NULL);
                        smap_add(&ppr, des, ppr_incre);
                    }
                }
            }
        }
        /* algo.h:578-590 kth_ppr: k-th largest over ppr.occur (0 if fewer than k entries;
         * the reference indexes out of range there). */
        double kth = 0;
        if (ppr.n_occur >= k) {
            for (int64_t i = 0; i < ppr.n_occur; i++) tmp[i] = ppr.val[ppr.occur[i]];
            qsort(tmp, (size_t)ppr.n_occur, sizeof(double), cmp_desc_d);
            kth = tmp[k - 1];
        }
        if (kth >= (1 + epsilon) * delta || delta <= min_delta) break; /* :1030 */
        delta = delta / 4.0 > min_delta ? delta / 4.0 : min_delta;     /* :1041 */
    }
    /* algo.h:592-610 topk_ppr: k (id, score) pairs, score descending, padded with (0, 0.0). */
    idsc *all = (idsc *)malloc(sizeof(idsc) * (size_t)(ppr.n_occur + 1));
    for (int64_t i = 0; i < ppr.n_occur; i++) { all[i].id = ppr.occur[i]; all[i].sc = ppr.val[ppr.occur[i]]; }
    qsort(all, (size_t)ppr.n_occur, sizeof(idsc), cmp_idsc);
    for (int32_t i = 0; i < k; i++) {
        if (i < ppr.n_occur) { ids[i] = all[i].id; scores[i] = all[i].sc; }
        else { ids[i] = 0; scores[i] = 0.0; }
    }
    if (rounds) *rounds = nround;
    if (ppr_out) memcpy(ppr_out, ppr.val, sizeof(double) * (size_t)n);
    free(all); free(tmp); free(f1); free(f2); free(rw_counter); free(forward_from.a);
    smap_free(&reserve); smap_free(&residue); smap_free(&ppr);
    return 0;
}

/* The pushes of the first `rounds` rounds of fora_query_topk_new (query.h:1001-1041) for one source, without the walks
 * between them: forward_local_update_linear_topk works on reserve / residue / forward_from only, so its work does not
 * depend on the walks -- only the NUMBER of rounds does (the stop test, :1030), and the caller passes the number its
 * own run took.  Returns the pops and edge relaxations of those pushes: the algorithmic counts of bench.py's roofline
 * entry for the top-k configuration. */
int orc_topk_push_counts(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s, int32_t k,
                         double epsilon, double alpha, double rmax_scale, int32_t rounds, int64_t *pops, int64_t *relax) {
    if (k == 0) k = 500;
    const double min_delta = 1.0 / n, init_delta = 1.0 / k / 10, pfail = 1.0 / n / n;
    const double lowest_delta_rmax = epsilon * sqrt(min_delta / 3 / m / log(2 / pfail));
    double delta = init_delta, rsum = 1.0;
    smap reserve, residue;
    smap_init(&reserve, n); smap_init(&residue, n);
    ivec forward_from = {0, 0, 0};
    ivec_push(&forward_from, s);
    smap_insert(&residue, s, rsum);
    unsigned char *f1 = (unsigned char *)malloc((size_t)n), *f2 = (unsigned char *)malloc((size_t)n);
    topk_push_pops = 0; topk_push_relax = 0;
    for (int32_t r = 0; r < rounds && delta >= min_delta; r++) {
        double rmax, omega;
        orc_fora_topk_setting(m, epsilon, delta, pfail, rmax_scale, &rmax, &omega);
        if (row_ptr[s + 1] == row_ptr[s]) break; /* :1007-1011 */
        push_fifo_topk(n, row_ptr, col, s, &rsum, rmax, lowest_delta_rmax, alpha, &reserve, &residue, &forward_from, f1, f2);
        if (delta <= min_delta) break;
        delta = delta / 4.0 > min_delta ? delta / 4.0 : min_delta;
    }
    if (pops) *pops = topk_push_pops;
    if (relax) *relax = topk_push_relax;
    free(f1); free(f2); free(forward_from.a);
    smap_free(&reserve); smap_free(&residue);
    return 0;
}

/* ---------------------------------------------------- top-k with bounds (non --opt) */
/* algo.h:1169-1174 calculate_lambda, operand order kept. */
double orc_calculate_lambda(double rsum, double pfail, double upper_bound, long total_rw_num) {
    return 1.0 / 3 * log(2 / pfail) * rsum / total_rw_num +
           sqrt(4.0 / 9.0 * log(2.0 / pfail) * log(2.0 / pfail) * rsum * rsum +
                8 * total_rw_num * log(2.0 / pfail) * rsum * upper_bound) /
               2.0 / total_rw_num;
}

/* algo.h:1178-1261 set_ppr_bounds.  upper / lower: dense n (init_keys + reset_one / reset_zero,
 * query.h:1350-1353, :939-940), so exist() is always true for them. */
static void set_ppr_bounds(int32_t n, double rsum, long total_rw_num, double pfail, const smap *reserve_m,
                           const smap *ppr, double *upper, double *lower) {
    const double min_ppr = 1.0 / n;
    const double sqrt_min_ppr = sqrt(1.0 / n);
    double epsilon_v_div = sqrt(2.67 * rsum * log(2.0 / pfail) / total_rw_num);
    double default_epsilon_v = epsilon_v_div / sqrt_min_ppr;
    for (int64_t i = 0; i < ppr->n_occur; i++) {
        int32_t nodeid = ppr->occur[i];
        if (ppr->val[nodeid] <= 0) continue;
        double reserve = 0.0;
        if (reserve_m->has[nodeid]) reserve = reserve_m->val[nodeid];
        double epsilon_a;
        if (upper[nodeid] > reserve) /* :1211-1215 (upper_bounds.exist is always true) */
            epsilon_a = orc_calculate_lambda(rsum, pfail, upper[nodeid] - reserve, total_rw_num);
        else
            epsilon_a = orc_calculate_lambda(rsum, pfail, 1 - reserve, total_rw_num);
        double ub_eps_a = ppr->val[nodeid] + epsilon_a;
        double lb_eps_a = ppr->val[nodeid] - epsilon_a;
        if (!(lb_eps_a > 0)) lb_eps_a = 0;
        double epsilon_v = default_epsilon_v;
        if (reserve_m->has[nodeid] && reserve_m->val[nodeid] > min_ppr) { /* :1230-1233 */
            reserve = reserve > lower[nodeid] ? reserve : lower[nodeid];
            epsilon_v = epsilon_v_div / sqrt(reserve);
        } else if (lower[nodeid] > 0) { /* :1235-1236 */
            epsilon_v = epsilon_v_div / sqrt(lower[nodeid]);
        }
        double ub_eps_v = 1.0, lb_eps_v = 0.0;
        if (1.0 - epsilon_v > 0) {
            ub_eps_v = ppr->val[nodeid] / (1.0 - epsilon_v);
            lb_eps_v = ppr->val[nodeid] / (1.0 + epsilon_v);
        }
        double up_bound = ub_eps_a < ub_eps_v ? ub_eps_a : ub_eps_v;
        if (!(up_bound < 1.0)) up_bound = 1.0;
        double low_bound = lb_eps_a > lb_eps_v ? lb_eps_a : lb_eps_v;
        if (!(low_bound > reserve)) low_bound = reserve;
        if (up_bound > 0) upper[nodeid] = up_bound;
        if (low_bound >= 0) lower[nodeid] = low_bound;
    }
}

typedef struct { int32_t id; double sc; } idlb;
static int cmp_idlb(const void *a, const void *b) {
    const idlb *x = (const idlb *)a, *y = (const idlb *)b;
    if (x->sc != y->sc) return (x->sc < y->sc) - (x->sc > y->sc);
    return (x->id > y->id) - (x->id < y->id); /* tie order unspecified in algo.h:1123; id asc here */
}

/* algo.h:1096-1166 if_stop. */
static int if_stop_bound(int32_t n, int32_t k, double delta, double threshold, double epsilon, const smap *ppr,
                         const double *upper, const double *lower, double *tmp, idlb *tb, unsigned char *filter) {
    double kth = 0; /* kth_ppr, algo.h:578-590 (0 when fewer than k entries: the reference reads out of range) */
    if (ppr->n_occur >= k) {
        for (int64_t i = 0; i < ppr->n_occur; i++) tmp[i] = ppr->val[ppr->occur[i]];
        qsort(tmp, (size_t)ppr->n_occur, sizeof(double), cmp_desc_d);
        kth = tmp[k - 1];
    }
    if (kth >= 2.0 * delta) return 1;
    if (delta >= threshold) return 0;
    const double error = 1.0 + epsilon, error_2 = 1.0 + epsilon;
    for (int32_t v = 0; v < n; v++) { tb[v].id = v; tb[v].sc = lower[v]; } /* lower_bounds.occur = 0..n-1 */
    qsort(tb, (size_t)n, sizeof(idlb), cmp_idlb);                          /* partial_sort_copy :1122 */
    memset(filter, 0, (size_t)n);
    for (int32_t i = 0; i < k; i++) { /* :1128-1136 */
        filter[tb[i].id] = 1;
        double ratio = upper[tb[i].id] / lower[tb[i].id];
        if (ratio > error_2) return 0;
    }
    double low_bound_k = tb[k - 1].sc;
    if (low_bound_k <= delta) return 0; /* :1145-1147 */
    for (int32_t v = 0; v < n; v++) {    /* :1148-1163; ppr[v] is nil (-9) when absent */
        if (filter[v] || !ppr->has[v] || ppr->val[v] <= 0) continue;
        double upper_temp = upper[v], lower_temp = lower[v];
        if (upper_temp > low_bound_k * error) {
            if (upper_temp > (1 + epsilon) / (1 - epsilon) * lower_temp) continue;
            else return 0;
        }
    }
    return 1;
}

/* query.h:909-969 fora_query_topk_with_bound + compute_ppr_with_fwdidx_topk_with_bound (query.h:639-750)
 * + topk_ppr (algo.h:592-610).  zero_ppr_upper_bound (query.h:935, :748) only ever feeds itself and is not
 * kept.  Walk numbering: walk j of node `source` in round `nround` (index entries first, then online). */
int orc_topk_bound_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                         int32_t k, double epsilon, double alpha, double rmax_scale, double ppr_decay_alpha,
                         uint64_t seed, const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                         int32_t *ids, double *scores, int32_t *rounds, double *ppr_out) {
    const double min_delta = 1.0 / n;                                                    /* :911 */
    const double init_delta = 1.0 / 4;                                                   /* :912 */
    const double threshold = (1.0 - ppr_decay_alpha) / pow(500, ppr_decay_alpha) / pow(n, 1 - ppr_decay_alpha); /* :913 */
    const double new_pfail = 1.0 / n / n / log(n);                                       /* :915 */
    double pfail = new_pfail, delta = init_delta;                                        /* :917-918 */
    const double lowest_delta_rmax = epsilon * sqrt(min_delta / 3 / m / log(2 / new_pfail)); /* :920 */
    double rsum = 1.0;
    smap reserve, residue, ppr;
    smap_init(&reserve, n); smap_init(&residue, n); smap_init(&ppr, n);
    ivec forward_from = {0, 0, 0};
    ivec_push(&forward_from, s);
    smap_insert(&residue, s, rsum);
    uint64_t *rw_counter = rw_idx ? (uint64_t *)calloc((size_t)n, sizeof(uint64_t)) : NULL; /* :937-938 */
    double *upper = (double *)malloc(sizeof(double) * (size_t)n), *lower = (double *)calloc((size_t)n, sizeof(double));
    for (int32_t v = 0; v < n; v++) upper[v] = 1.0; /* :941-942 */
    unsigned char *f1 = (unsigned char *)malloc((size_t)n), *f2 = (unsigned char *)malloc((size_t)n);
    unsigned char *filter = (unsigned char *)malloc((size_t)n);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    idlb *tb = (idlb *)malloc(sizeof(idlb) * (size_t)n);
    int32_t nround = 0;

    while (delta >= min_delta) { /* :944 */
        double rmax = epsilon * sqrt(delta / 3 / m / log(2 / pfail)); /* fora_setting, algo.h:455-463 */
        rmax *= rmax_scale;
        double omega = (2 + epsilon) * log(2 / pfail) / delta / epsilon / epsilon;
        nround++;
        if (row_ptr[s + 1] == row_ptr[s]) { /* :951-955 */
            rsum = 0.0;
            smap_insert(&reserve, s, 1);
            smap_clean(&ppr);
            smap_insert(&ppr, s, 1);
            break;
        }
        push_fifo_topk(n, row_ptr, col, s, &rsum, rmax, lowest_delta_rmax, alpha, &reserve, &residue,
                       &forward_from, f1, f2); /* :957 */
        /* compute_ppr_with_fwdidx_topk_with_bound */
        smap_clean(&ppr);
        for (int64_t i = 0; i < reserve.n_occur; i++) {
            int32_t v = reserve.occur[i];
            if (reserve.val[v]) smap_insert(&ppr, v, reserve.val[v]);
        }
        if (rsum != 0.0) { /* :642-643 */
            long num_random_walk = (long)(omega * rsum); /* :645 */
            long real_num_rand_walk = 0;
            if (rw_idx) { /* :652-721 */
                qsort(residue.occur, (size_t)residue.n_occur, sizeof(int32_t), cmp_i32);
                for (int64_t i = 0; i < residue.n_occur; i++) {
                    int32_t source = residue.occur[i];
                    double residual = residue.val[source];
                    long num_s_rw = (long)ceil(residual * omega);                         /* :659 */
                    double a_s = residual / rsum * num_random_walk / num_s_rw;             /* :660 */
                    double ppr_incre = a_s * rsum / num_random_walk;                       /* :662 */
                    real_num_rand_walk += num_s_rw;
                    uint64_t used = rw_counter[source];
                    long remaining = (long)(cnt[source] - used);
                    long from_idx = num_s_rw <= remaining ? num_s_rw : remaining;
                    for (long kk = 0; kk < from_idx; kk++) smap_add(&ppr, rw_idx[off[source] + used + (uint64_t)kk], ppr_incre);
                    rw_counter[source] = used + (uint64_t)from_idx;
                    for (long j = from_idx; j < num_s_rw; j++) { /* :711-718 */
                        int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, (uint32_t)nround, source,
                                               (uint64_t)j, alpha, 0, NULL);
                        smap_add(&ppr, des, ppr_incre);
                    }
                }
            } else { /* :723-741 */
                for (int64_t i = 0; i < residue.n_occur; i++) {
                    int32_t source = residue.occur[i];
                    double residual = residue.val[source];
                    long num_s_rw = (long)ceil(residual / rsum * num_random_walk);         /* :727 */
                    double a_s = residual / rsum * num_random_walk / num_s_rw;
                    real_num_rand_walk += num_s_rw;
                    double ppr_incre = a_s * rsum / num_random_walk;
                    for (long j = 0; j < num_s_rw; j++) {
                        int32_t des = orc_walk(n, row_ptr, col, seed, (uint32_t)s, (uint32_t)nround, source,
                                               (uint64_t)j, alpha, 0, NULL);
                        smap_add(&ppr, des, ppr_incre);
                    }
                }
            }
            if (delta < threshold) /* :745-746 */
                set_ppr_bounds(n, rsum, real_num_rand_walk, pfail, &reserve, &ppr, upper, lower);
        }
        if (if_stop_bound(n, k, delta, threshold, epsilon, &ppr, upper, lower, tmp, tb, filter) || delta <= min_delta)
            break; /* :962-964 */
        delta = delta / 2.0 > min_delta ? delta / 2.0 : min_delta; /* :966 */
    }
    idsc *all = (idsc *)malloc(sizeof(idsc) * (size_t)(ppr.n_occur + 1));
    for (int64_t i = 0; i < ppr.n_occur; i++) { all[i].id = ppr.occur[i]; all[i].sc = ppr.val[ppr.occur[i]]; }
    qsort(all, (size_t)ppr.n_occur, sizeof(idsc), cmp_idsc);
    for (int32_t i = 0; i < k; i++) {
        if (i < ppr.n_occur) { ids[i] = all[i].id; scores[i] = all[i].sc; }
        else { ids[i] = 0; scores[i] = 0.0; }
    }
    if (rounds) *rounds = nround;
    if (ppr_out) memcpy(ppr_out, ppr.val, sizeof(double) * (size_t)n);
    free(all); free(tmp); free(tb); free(f1); free(f2); free(filter); free(rw_counter); free(forward_from.a);
    free(upper); free(lower);
    smap_free(&reserve); smap_free(&residue); smap_free(&ppr);
    return 0;
}

/* ---------------------------------------------------------------- exact PPR */
/* query.h:1192-1224 fwd_power_iteration, dense: alpha*r kept, (1-alpha)*r spread over
 * out-neighbours, dangling mass returned to the start node (:1210-1212). */
void orc_power_iteration(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s,
                         double alpha, int iters, double *ppr) {
    double *r = (double *)calloc((size_t)n, sizeof(double));
    double *r2 = (double *)calloc((size_t)n, sizeof(double));
    memset(ppr, 0, sizeof(double) * (size_t)n);
    r[s] = 1.0;
    for (int it = 0; it < iters; it++) {
        memset(r2, 0, sizeof(double) * (size_t)n);
        for (int32_t v = 0; v < n; v++) {
            double p = r[v];
            if (p > 0) {
                ppr[v] += alpha * p;
                int64_t deg = row_ptr[v + 1] - row_ptr[v];
                double remain = (1 - alpha) * p;
                if (deg == 0)
                    r2[s] += remain;
                else {
                    double avg = remain / deg;
                    for (int64_t e = row_ptr[v]; e < row_ptr[v + 1]; e++) r2[col[e]] += avg;
                }
            }
        }
        double *t = r; r = r2; r2 = t;
    }
    free(r); free(r2);
}
