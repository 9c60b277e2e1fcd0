/*
 * fora_oracle.h -- CPU restatement of the FORA SSPPR hot path of wangsibovictor/fora.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fora_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU baseline.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors for this path
 * and cannot be built in this image (every TU needs Boost, which is absent;
 * stand-in headers are not allowed).  What IS pinned: the reference's own data
 * fixtures (data/webstanford/attribute.txt, ssquery.txt -> tests/golden/), the
 * Random123 known-answer vectors for Philox4x32-10, and the mathematical
 * definition of PPR through the power iteration of query.h:1192-1224.
 *
 * Two families of functions live here:
 *   orc_*       the reference algorithm in the reference's own arithmetic
 *               (f64, FIFO order), each citing the file:line it follows.
 *   orc_twin_*  the same algorithm in the level-synchronous schedule and
 *               2^-62 fixed-point arithmetic the HIP path uses, so that the
 *               HIP path can be checked BIT-EXACTLY (integer atomics are
 *               order independent).  tests/ tie the two families together
 *               (same invariants, same epsilon guarantee, stated L-inf gap).
 *
 * Citations are into /root/reference.
 */
#ifndef FORA_ORACLE_H
#define FORA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FIX_ONE (1ULL << 62)
#define ORC_STREAM_INDEX 0xFFFFFFFFu /* Philox counter word 3 for index walks */

/* ---- graph loading: graph.h:48-64 (init_nm), graph.h:151-161 (plain branch) ---- */
int orc_read_attribute(const char *path, int32_t *n, int64_t *m);
/* counts edges in a "src dst" text file (pairs read with %d%d like fscanf) */
int64_t orc_count_edges(const char *path);
int orc_read_edges(const char *path, int32_t *src, int32_t *dst, int64_t cap, int64_t *ne);
/* CSR with per-row FILE ORDER kept, self loops dropped, duplicates kept
 * (graph.h:157-158).  Returns nnz, or -1 if an id >= n (assert graph.h:155-156).
 * row_ptr has n+1 entries; col must hold ne entries. */
int64_t orc_build_csr(int32_t n, const int32_t *src, const int32_t *dst, int64_t ne,
                      int64_t *row_ptr, int32_t *col);
int64_t orc_read_queries(const char *path, int32_t *out, int64_t cap); /* algo.h:511-522 */

/* ---- parameters: graph.h:173-183, algo.h:455-463, algo.h:466-474 ---- */
void orc_fora_setting(int32_t n, int64_t m, double epsilon, double alpha, double rmax_scale,
                      int opt, double *rmax, double *omega);
void orc_fora_topk_setting(int64_t m, double epsilon, double delta, double pfail,
                           double rmax_scale, double *rmax, double *omega);

/* ---- Philox4x32-10 and the walk contract (replaces Boost taus88 / lagged_fibonacci607,
 *      algo.h:105-122; walk semantics algo.h:124-166) ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int32_t orc_walk(int32_t n, const int64_t *row_ptr, const int32_t *col, uint64_t seed,
                 uint32_t stream, uint32_t round, int32_t start, uint64_t j, double alpha,
                 int no_zero_hop, int64_t *steps);

/* ---- FIFO forward push, f64: algo.h:954-1018 ---- */
typedef struct {
    double rsum;
    int64_t pops;        /* queue pops (P of SURVEY 8d) */
    int64_t relax;       /* edge relaxations (E of SURVEY 8d) */
    int64_t n_reserve;   /* entries in reserve_occur */
    int64_t n_residue;   /* entries in residue_occur */
    int64_t generations; /* FIFO generations */
} orc_push_stats;
/* reserve/residue: n doubles, filled with nil=-1 first like iMap::initialize
 * (query.h:1464-1467); *_occur: n int32 each (first-touch order, mylib.h:387-399). */
int orc_push_fifo(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                  double alpha, double *reserve, double *residue, int32_t *reserve_occur,
                  int32_t *residue_occur, orc_push_stats *st);

/* ---- walk allocation: query.h:270,282-285,314-317 (opt: query.h:349,363-367) ----
 * For every entry of residue_occur writes num_s_rw; returns N = (u64)(omega*rsum'). */
uint64_t orc_walk_counts(const double *residue, const int32_t *residue_occur, int64_t n_residue,
                         double rsum, double omega, double alpha, int opt, uint64_t *num_s_rw);

/* ---- walk index: build.h:302-366 (sizes :325-334, walks :344-354) ---- */
uint64_t orc_index_sizes(int32_t n, const int64_t *row_ptr, double rmax, double omega, double alpha,
                         int opt, uint64_t *off, uint64_t *cnt);
void orc_build_index(int32_t n, const int64_t *row_ptr, const int32_t *col, uint64_t seed,
                     double alpha, int opt, const uint64_t *off, const uint64_t *cnt,
                     int32_t *rw_idx);

/* ---- refinement: query.h:255-327 (plain), query.h:334-413 (--opt) ----
 * ppr: n doubles (dense, query.h:1427).  rw_idx==NULL -> online walks. */
typedef struct {
    uint64_t n_walks;    /* num_total_rw */
    uint64_t n_idx_hit;  /* num_hit_idx */
    uint64_t walk_steps; /* online steps taken */
} orc_refine_stats;
int orc_refine(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s,
               const double *reserve, const int32_t *reserve_occur, int64_t n_reserve,
               const double *residue, int32_t *residue_occur, int64_t n_residue, double rsum,
               double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
               const uint64_t *off, const uint64_t *cnt, double *ppr, orc_refine_stats *st);

/* ---- fora_query_basic: query.h:841-907 (non --balanced) ---- */
/* the query loop on `threads` host threads for `seconds` at most (bench.py's all-core baseline): queries finished */
int64_t orc_query_many(int32_t n, const int64_t *row_ptr, const int32_t *col, const int32_t *sources, int64_t nsrc,
                       double rmax, double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                       const uint64_t *off, const uint64_t *cnt, int threads, double seconds, double *elapsed,
                       uint64_t *walks);
/* the same with thread t bound to host CPU cpus[t mod ncpus] (bench.py: one thread per CPU the process may run on) */
int64_t orc_query_many_pinned(int32_t n, const int64_t *row_ptr, const int32_t *col, const int32_t *sources, int64_t nsrc,
                              double rmax, double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                              const uint64_t *off, const uint64_t *cnt, int threads, double seconds, double *elapsed,
                              uint64_t *walks, const int32_t *cpus, int ncpus);
int orc_query(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
              double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
              const uint64_t *off, const uint64_t *cnt, double *ppr, orc_push_stats *pst,
              orc_refine_stats *rst);

/* ---- top-k (--opt driver): query.h:972-1045, algo.h:1020-1093, query.h:521-636,
 *      query.h:243-253, algo.h:578-610 ---- */
int orc_topk_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                   int32_t k, double epsilon, double alpha, double rmax_scale, uint64_t seed,
                   const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                   int32_t *ids, double *scores, int32_t *rounds, double *ppr_out);

/* Pops and edge relaxations of the pushes of the first `rounds` rounds of the --opt top-k driver for one source
 * (algo.h:1020-1093 under query.h:1001-1041), without the walks between them. */
int orc_topk_push_counts(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s, int32_t k,
                         double epsilon, double alpha, double rmax_scale, int32_t rounds, int64_t *pops, int64_t *relax);

/* ---- top-k with bounds (get_topk without --opt): query.h:909-969, :639-750, algo.h:1096-1261 ---- */
double orc_calculate_lambda(double rsum, double pfail, double upper_bound, long total_rw_num);
int orc_topk_bound_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                         int32_t k, double epsilon, double alpha, double rmax_scale, double ppr_decay_alpha,
                         uint64_t seed, const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                         int32_t *ids, double *scores, int32_t *rounds, double *ppr_out);

/* ---- exact PPR: query.h:1192-1224 (dense restatement, `iters` rounds) ---- */
void orc_power_iteration(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s,
                         double alpha, int iters, double *ppr);

/* ======================= schedule twin of the HIP path ======================= */
uint64_t orc_twin_alpha_fix(double alpha);
uint64_t orc_twin_rmax_fix(double rmax);
typedef struct {
    uint64_t rsum_fix;
    int64_t levels;
    int64_t pops;
    int64_t relax;
} orc_twin_push_stats;
/* residue, ppr: n u64, zeroed by the callee.  level_sizes (optional, cap entries). */
/* threshold rounds of orc_twin_push / orc_twin_query (the engine's option "rounds"; default 1) */
void orc_twin_set_defer(int k); /* bounded deferral of the push (the engine's option "defer"; default 0: plain levels) */
int orc_twin_get_defer(void);
void orc_twin_set_defer_min(int64_t m); /* with defer: only levels that pop at least m nodes defer (option "defer_min"; default 0) */
void orc_twin_set_rounds(int rounds);
int orc_twin_get_rounds(void);
void orc_twin_set_round_div(int div); /* 0 (default): a round ends when its frontier is empty */
int orc_twin_get_round_div(void);
int orc_twin_push(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                  double alpha, uint64_t *residue, uint64_t *ppr, orc_twin_push_stats *st,
                  int64_t *level_sizes, int64_t cap);
/* fwd_power_iteration (query.h:1192-1224) in the twin's arithmetic; ppr: n u64 */
int orc_twin_power_iteration(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double alpha,
                             int32_t max_iter, uint64_t *ppr, orc_twin_push_stats *st);
uint64_t orc_twin_walk_counts(int32_t n, const uint64_t *residue, uint64_t rsum_fix, double omega,
                              double alpha, int opt, uint64_t *num_s_rw);
int orc_twin_refine(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s,
                    const uint64_t *residue, uint64_t rsum_fix, double omega, double alpha, int opt,
                    uint64_t seed, const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                    uint64_t *ppr, orc_refine_stats *st);
int orc_twin_query(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax,
                   double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                   const uint64_t *off, const uint64_t *cnt, uint64_t *residue, uint64_t *ppr,
                   orc_twin_push_stats *pst, orc_refine_stats *rst);
int orc_twin_topk_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                        int32_t k, double epsilon, double alpha, double rmax_scale, uint64_t seed,
                        const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                        int32_t *ids, double *scores, int32_t *rounds, uint64_t *ppr_out);
/* top-k with bounds in the twin's schedule; upper_out / lower_out (n doubles, optional): final bounds */
int orc_twin_topk_bound_query(int32_t n, int64_t m, const int64_t *row_ptr, const int32_t *col, int32_t s,
                              int32_t k, double epsilon, double alpha, double rmax_scale, double ppr_decay_alpha,
                              uint64_t seed, const int32_t *rw_idx, const uint64_t *off, const uint64_t *cnt,
                              int32_t *ids, double *scores, int32_t *rounds, uint64_t *ppr_out,
                              double *upper_out, double *lower_out);
/* --balanced (query.h:848-884) with the push charged by its work counters; returns the rounds run */
int orc_twin_query_balanced(int32_t n, const int64_t *row_ptr, const int32_t *col, int32_t s, double rmax0,
                            double omega, double alpha, int opt, uint64_t seed, const int32_t *rw_idx,
                            const uint64_t *off, const uint64_t *cnt, double start_scale, double c_pop, double c_edge,
                            double t_walk, double t_idx, uint64_t *residue, uint64_t *ppr, orc_twin_push_stats *pst,
                            orc_refine_stats *rst, double *rmax_out);
void orc_twin_bounds_node(double p, double reserve, double rsum, double L, double total, double min_ppr,
                          double sqrt_min_ppr, double *upper, double *lower);
void orc_fix_to_double(const uint64_t *in, int64_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif
