/*
 * fora_hip.h -- C ABI of the MI355X-native FORA SSPPR engine (libfora_hip.so).
 *
 * The reference (wangsibovictor/fora) exposes no plugin / FFI interface; its
 * seams are C++ free functions that read and write global state from one thread
 * (SURVEY.md 8b).  Each entry point below names the reference seam it replaces.
 * Conventions: plain C types only, caller-allocated outputs, return 0 on success
 * or a negative FORA_E_* code (never throws, never exit()s), one ctx per GPU, a
 * ctx is used by one host thread at a time.  All host pointers are ordinary
 * pageable memory; the library copies.
 *
 * NUMERIC CONTRACT.  residue / reserve / ppr live on the device as unsigned 2^-62 fixed point
 * (1.0 == FORA_FIX_ONE), so every accumulation is an exact, order-independent integer add:
 *   - results are bit-reproducible (same inputs, same seed -> same bits, whatever the batch size, the bucket
 *     layout or the GPU count) and equal oracle/fora_twin.c bit for bit;
 *   - mass is conserved EXACTLY: after the push sum(reserve) + sum(residue) == FORA_FIX_ONE, after the refinement
 *     sum(ppr) == FORA_FIX_ONE (fora_query_stats.ppr_sum_fix); walk j of a residue node carries floor(r / num_s_rw)
 *     plus one more unit for j < r mod num_s_rw;
 *   - a pop keeps floor(alpha * r) -- alpha enters as floor(alpha * 2^62) -- and every out-edge gets
 *     floor((r - keep) / outdeg); the division remainder (< outdeg units of 2^-62 = 2.2e-19) stays in the RESERVE of
 *     the popped node, where the reference's f64 `((1-alpha)*r)/outdeg` (algo.h:1002) rounds instead;
 *   - the threshold test residue/outdeg >= rmax (algo.h:1012) is residue >= ceil(rmax * 2^62) * outdeg;
 *   - double-typed outputs (ppr_out, scores, rsum) are value * 2^-62: absolute resolution 2.2e-19, i.e. at least
 *     9 significant digits for any entry >= 1/n of a graph of up to 2^31 nodes.
 * ERROR BOUND of ppr_out against exact PPR pi(s, .) (power iteration, query.h:1192-1224): the FORA guarantee the
 * parameters of algo.h:455-463 encode, |ppr - pi| <= epsilon * pi for every pi >= 1/n with probability
 * 1 - 1/n; the fixed point adds at most (pops + relaxations + walks) * 2^-62 < 1e-8 * that bound.  Checked at
 * n = 281 904 in tests/test_hip_parity_gpu.py::test_full_size_webstanford_properties (L-inf <= 2e-5).
 * RNG CONTRACT, version 2 (replaces the time(0)-seeded Boost engines of algo.h:105-122): Philox4x32-10, key = seed,
 * one call per two steps, counter = (start node, walk# lo32, walk# bits 32..47 | round << 16 | (step/2 & 255) << 24,
 * stream ^ (step >> 9) * 0x9E3779B9), stream = the query's source id (FORA_STREAM_INDEX for index walks).
 * Version 1 (round 1) lacked the (step >> 9) term; walks of up to 512 steps are identical in both.
 */
#ifndef FORA_HIP_H
#define FORA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FORA_FIX_ONE (1ULL << 62)
#define FORA_STREAM_INDEX 0xFFFFFFFFu

enum {
    FORA_OK = 0,
    FORA_E_ARG = -1,      /* bad argument / call order (reference: assert, graph.h:155) */
    FORA_E_HIP = -2,      /* HIP runtime error, see fora_hip_last_error */
    FORA_E_NOMEM = -3,    /* device memory */
    FORA_E_OVERFLOW = -4, /* internal work list overflow / level cap reached */
    FORA_E_NOGPU = -5     /* no usable gfx950 device */
};

typedef struct fora_ctx fora_ctx;

/* Per-query counters.  Replaces the globals fora_query_basic updates:
 * num_total_rw / num_hit_idx (algo.h:39-40), rsum (query.h:843), plus schedule
 * counters of the level-synchronous push. */
typedef struct {
    double rsum;         /* residue mass left by the push, = rsum_fix * 2^-62 (query.h:843,886) */
    uint64_t rsum_fix;
    uint64_t n_rw;       /* N = (u64)(omega*rsum') of query.h:270 */
    uint64_t n_walks;    /* sum of num_s_rw, query.h:287,318 (num_total_rw) */
    uint64_t n_idx_hit;  /* walks served from the index, query.h:295,306 (num_hit_idx) */
    uint64_t pops;       /* frontier pops of the push */
    uint64_t relax;      /* edge relaxations of the push */
    uint64_t ppr_sum_fix; /* sum of the final ppr vector (== FORA_FIX_ONE when mass is conserved) */
    int32_t levels;      /* push levels this query was active in */
    int32_t dangling_source; /* 1: algo.h:961-965 fast path taken */
    double rmax_used;    /* rmax of the last push round (config.rmax unless --balanced, query.h:877) */
    int32_t push_rounds; /* 1 unless --balanced */
    int32_t reserved_;
} fora_query_stats;

/* Accumulated device timings since the last reset (HIP events on the ctx stream). */
typedef struct {
    double push_pop_ms;     /* sum over k_push_pop launches (direct path only) */
    double push_expand_ms;  /* sum over k_pushq_bin launches (k_push_expand on the direct path) */
    double push_accum_ms;   /* sum over k_accum<false> launches (bucketed push only) */
    double walk_alloc_ms;   /* k_walk_alloc */
    double walk_ms;         /* k_walk_idx + k_walk_online */
    double walk_accum_ms;   /* k_accum<to ppr> (bucketed path only) */
    double other_ms;        /* init / reduce / convert kernels + memsets */
    double batch_ms;        /* whole batches, first launch to last completion */
    uint64_t push_pop_launches;
    uint64_t push_expand_launches;
    uint64_t push_accum_launches;
    uint64_t walk_launches;
    uint64_t batches;
    uint64_t pops;          /* totals over all queries since reset */
    uint64_t relax;
    uint64_t walks;
    uint64_t walk_steps;
    uint64_t levels;        /* levels launched (including speculative empty ones) */
    uint64_t idx_hits;      /* walks served from the index (num_hit_idx, algo.h:39) */
    double push_tail_ms;    /* k_push_tail launches (they finish the push once every frontier is small) */
    uint64_t push_tail_launches;
    double push_team_ms;    /* k_push_team launches (graphs of the narrow layout: the whole push above the tail, one launch per batch) */
    uint64_t push_team_launches;
} fora_timing;

/* ---- lifecycle ---------------------------------------------------------- */
int fora_hip_device_count(void); /* usable HIP devices (0 when there is none) */
int fora_hip_create(int device, fora_ctx **out);
void fora_hip_destroy(fora_ctx *ctx);
const char *fora_hip_last_error(fora_ctx *ctx);
/* device name / arch string of the ctx's GPU, e.g. "gfx950:sramecc+:xnack-" */
int fora_hip_device_info(fora_ctx *ctx, char *arch, int arch_len, int *cus, uint64_t *hbm_bytes);

/* ---- graph: replaces Graph::init_graph's in-memory result (graph.h:89-163) --
 * CSR with per-row file order; m_attr is the m of attribute.txt (graph.h:58-63),
 * which the parameter formulas use even when it differs from nnz. */
int fora_hip_set_graph(fora_ctx *ctx, int32_t n, int64_t m_attr, const int64_t *row_ptr,
                       const int32_t *col);

/* ---- parameters: replaces init_parameter (graph.h:173-183) + fora_setting
 * (algo.h:455-463); alpha is config.alpha (config.h:27,132). */
int fora_hip_set_params(fora_ctx *ctx, double alpha, double epsilon, double rmax_scale, int opt,
                        uint64_t seed);
/* same, with rmax / omega given directly (tests, --balanced style callers) */
int fora_hip_set_params_raw(fora_ctx *ctx, double alpha, double rmax, double omega, int opt,
                            uint64_t seed);
int fora_hip_get_params(fora_ctx *ctx, double *rmax, double *omega);
/* --balanced (config.balanced; fora_query_basic query.h:848-884, estimated_random_walk_cost :825-838): rmax is
 * halved from 8*rmax while the estimated walk cost omega*rsum*(1-alpha)*t exceeds what the push has cost so far.
 * The reference measures the push with a wall clock (not reproducible); here it is charged by its work counters,
 * pops*c_pop + relax*c_edge seconds, and t = t_walk (t_idx with an index once rmax < config.rmax).  A cost <= 0
 * selects the value calibrated on MI355X (2.0e-11, 2.4e-11, 6.5e-11, 2.2e-11 s: twice the bulk push rate, because
 * the deeper rounds are long cascades of small levels; the reference's constants are
 * t_walk = 4e-7, t_idx = t_walk/140, query.h:822-823).  start_scale: first rmax = start_scale * config.rmax
 * (<= 0: the reference's 8, query.h:862; every round is a whole cascade of levels on the GPU, so callers after
 * throughput start at 1). */
int fora_hip_set_balanced(fora_ctx *ctx, int on, double start_scale, double c_pop, double c_edge, double t_walk,
                          double t_idx);
/* queries processed concurrently per launch; 0 = choose from free HBM */
int fora_hip_set_batch(fora_ctx *ctx, int batch);
int fora_hip_get_batch(fora_ctx *ctx);
/* Engine knobs (layout choice, launch shapes, capacities; the list is `OPTIONS` in fora_hip.hip).  A knob is read
 * once, in fora_hip_create, from the environment variable FORA_HIP_<NAME>; this call changes one afterwards (tests
 * use it to force the wide layout, tiny buckets, the k_push_tail or k_push_team path ...).  None of the knobs the
 * environment can set changes a result bit.  Four knobs choose another push SCHEDULE and with it other (equally valid)
 * residue / reserve / ppr bits: "rounds", "round_div" (threshold rounds), "defer" and "defer_min" (bounded deferral).
 * They are off by default, are NOT read from the environment -- only this call sets them -- and a run with them equals
 * oracle/fora_twin.c run with the same values (orc_twin_set_rounds / _round_div / _defer / _defer_min).
 * name "reset": back to the defaults / the environment values fora_hip_create read.  Unknown name: FORA_E_ARG. */
int fora_hip_set_option(fora_ctx *ctx, const char *name, int64_t value);
/* Reads a knob back, or one of the engine's read-only state words: "team_members" (members per team of k_push_team; 0:
 * this graph / workspace pushes with the bucketed kernels), "team_cooperative" (1: k_push_team is launched with
 * hipLaunchCooperativeKernel -- option team_coop, off by default), "team_fallbacks" (calls that were run again through the bucketed push because a team of
 * k_push_team timed out waiting for a member -- its workgroups were not co-resident, e.g. another context's kernels held
 * CUs; the caller sees FORA_OK and the same result bits), "team_suspended" (calls left that do not try the team push
 * after such a time-out), "bucket_retries" (calls that were run again with doubled message buckets), "test_paths" (1: the build
 * with the schedule experiments compiled in, libfora_hip_test.so). */
int fora_hip_get_option(fora_ctx *ctx, const char *name, int64_t *value);

/* ---- walk index: replaces build() (build.h:302-366), rw_idx / rw_idx_info
 * (algo.h:42-43) and deserialize_idx() (build.h:194-207) ------------------- */
int fora_hip_index_sizes(fora_ctx *ctx, uint64_t *total, uint64_t *off /*n or NULL*/,
                         uint64_t *cnt /*n or NULL*/);
int fora_hip_build_index(fora_ctx *ctx); /* walks on the GPU, index stays in HBM */
int fora_hip_get_index(fora_ctx *ctx, int32_t *rw_idx, uint64_t len, uint64_t *off, uint64_t *cnt);
int fora_hip_set_index(fora_ctx *ctx, const int32_t *rw_idx, uint64_t len, const uint64_t *off,
                       const uint64_t *cnt);
int fora_hip_clear_index(fora_ctx *ctx);

/* ---- SSPPR: replaces the query() loop over fora_query_basic
 * (query.h:1471-1476 -> query.h:841-907).  ppr_out: nq*n doubles or NULL
 * (results then stay in HBM; only stats come back).  with_idx: config.with_rw_idx. */
int fora_hip_query_batch(fora_ctx *ctx, const int32_t *sources, int nq, int with_idx,
                         double *ppr_out, fora_query_stats *stats /*nq or NULL*/);
/* same run, raw fixed-point outputs for bit-exact checks (either may be NULL) */
int fora_hip_query_batch_fix(fora_ctx *ctx, const int32_t *sources, int nq, int with_idx,
                             uint64_t *ppr_fix_out, uint64_t *residue_fix_out,
                             fora_query_stats *stats);

/* ---- top-k: replaces the topk() loop over get_topk -> fora_query_topk_new +
 * topk_ppr (query.h:1397-1401, 1139-1156, 972-1045; algo.h:592-610), --opt driver.
 * ids / scores: nq*k, score descending (ties: id ascending), padded with (0, 0.0). */
int fora_hip_topk_batch(fora_ctx *ctx, const int32_t *sources, int nq, int k, double epsilon,
                        double rmax_scale, int with_idx, int32_t *ids, double *scores,
                        int32_t *rounds /*nq or NULL*/);

/* ---- top-k with bounds: get_topk without --opt -> fora_query_topk_with_bound + topk_ppr (query.h:1139-1156,
 * 909-969, 639-750; set_ppr_bounds algo.h:1178-1261; if_stop algo.h:1096-1166).  ppr_decay_alpha: config.h:123
 * (0.77).  With with_idx the index must have been built without --opt.  Outputs as fora_hip_topk_batch. */
int fora_hip_topk_bound_batch(fora_ctx *ctx, const int32_t *sources, int nq, int k, double epsilon,
                              double rmax_scale, double ppr_decay_alpha, int with_idx, int32_t *ids,
                              double *scores, int32_t *rounds /*nq or NULL*/);

/* ---- exact SSPPR: replaces fwd_power_iteration / multi_power_iter under gen_exact_topk
 * (query.h:1192-1238, 1240-1307): max_iter Jacobi iterations (config.max_iter_num, config.h:115 = 100) of
 * "keep alpha of every positive residual, push the rest along the out-edges, dangling mass back to the
 * source", run as the level-synchronous push with the smallest threshold (one 2^-62 unit per out-edge).
 * Any of ppr_out (nq*n doubles), ppr_fix_out (nq*n raw) and ids/scores (nq*k, k >= 1, score descending,
 * ties id ascending, padded with (0, 0.0)) may be NULL. */
int fora_hip_power_iteration_batch(fora_ctx *ctx, const int32_t *sources, int nq, int max_iter,
                                   double *ppr_out, uint64_t *ppr_fix_out, int k, int32_t *ids,
                                   double *scores);

/* ---- stage hooks (same device code as the paths above, exposed for parity tests) */
/* forward push only (forward_local_update_linear, algo.h:954-1018) */
int fora_hip_push_batch(fora_ctx *ctx, const int32_t *sources, int nq, uint64_t *reserve_fix_out,
                        uint64_t *residue_fix_out, fora_query_stats *stats);
/* walk allocation in the reference's f64 arithmetic (query.h:270,282 / :349,:364):
 * residue: n doubles (<= 0 entries get 0 walks); returns N and num_s_rw[n]. */
int fora_hip_walk_counts(fora_ctx *ctx, const double *residue, double rsum, uint64_t *num_s_rw,
                         uint64_t *n_rw);
/* endpoints of `count` walks under the Philox contract (random_walk /
 * random_walk_no_zero_hop, algo.h:124-166) */
int fora_hip_walks(fora_ctx *ctx, uint32_t stream, uint32_t round, int no_zero_hop,
                   const int32_t *starts, const uint64_t *js, int64_t count, int32_t *dests);

/* ---- measurement ----------------------------------------------------------- */
int fora_hip_reset_timing(fora_ctx *ctx);
int fora_hip_get_timing(fora_ctx *ctx, fora_timing *out);
/* diagnostic builds (-DFORA_STAMPS) only: shader-clock cycles per kernel phase summed over workgroups since the last
 * fora_hip_reset_timing; [0..15] k_pushq_bin, [16..31] k_accum.  All zero in a product build. */
int fora_hip_get_stamps(fora_ctx *ctx, uint64_t *out32);

#ifdef __cplusplus
}
#endif
#endif
