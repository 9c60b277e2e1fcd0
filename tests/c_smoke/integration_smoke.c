/*
 * integration_smoke.c -- the reference-side binding of INTEGRATION.md, compiled as plain C against
 * include/fora_hip.h and linked with -lfora_hip: the calls a maintainer adds at fora.cpp:176-190 (attach),
 * query.h:1471-1476 (query loop), query.h:1397-1401 (top-k loop), build.h:344-354 (index build) and
 * query.h:1282-1290 (exact PPR), in that order, on a small synthetic graph.
 *
 * exit 0: every call succeeded and the results hold the contract stated in the header (mass 1 exactly,
 * sorted top-k, 100 % index hit); exit 77: no gfx950 device (FORA_E_NOGPU), nothing else was tried.
 */
#include "fora_hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != FORA_OK) {                                                               \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fora_hip_last_error(ctx));        \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)
#define EXPECT(cond)                                                                        \
    do { if (!(cond)) { fprintf(stderr, "expectation failed: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main(void) {
    /* a graph as Graph::init_graph leaves it (graph.h:151-161): n = 3000, every node 1..7 out-edges, file order */
    enum { N = 3000, NQ = 6, K = 20 };
    static int64_t row_ptr[N + 1];
    static int32_t col[N * 8];
    uint64_t x = 88172645463325252ULL;
    int64_t nnz = 0;
    for (int v = 0; v < N; v++) {
        row_ptr[v] = nnz;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        int deg = 1 + (int)(x % 7);
        for (int e = 0; e < deg; e++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            int w = (int)(x % N);
            if (w == v) w = (w + 1) % N; /* the loader drops self loops (graph.h:157) */
            col[nnz++] = w;
        }
    }
    row_ptr[N] = nnz;

    fora_ctx *ctx = NULL;
    int rc = fora_hip_create(0, &ctx);
    if (rc == FORA_E_NOGPU) { fprintf(stderr, "no gfx950 device\n"); return 77; }
    if (rc != FORA_OK) { fprintf(stderr, "fora_hip_create -> %d\n", rc); return 1; }

    /* ---- attach */
    double rmax = 0, omega = 0;
    CHECK(fora_hip_set_graph(ctx, N, nnz, row_ptr, col));
    CHECK(fora_hip_set_params(ctx, 0.2, 0.5, 1.0, /*opt*/ 0, /*seed*/ 0x464F5241ULL));
    CHECK(fora_hip_get_params(ctx, &rmax, &omega));
    {   /* fora_setting, algo.h:455-463, in the reference's operand order */
        double delta = 1.0 / N, pfail = 1.0 / N;
        double want_rmax = 0.5 * sqrt(delta / 3 / nnz / log(2 / pfail)) * 1.0;
        double want_omega = (2 + 0.5) * log(2 / pfail) / delta / 0.5 / 0.5;
        EXPECT(rmax == want_rmax && omega == want_omega);
    }

    /* ---- the query() loop */
    int32_t queries[NQ] = {5, 77, 1234, 2999, 0, 1500};
    fora_query_stats st[NQ];
    static double ppr[N];
    CHECK(fora_hip_query_batch(ctx, queries, NQ, /*with_idx*/ 0, NULL, st));
    for (int i = 0; i < NQ; i++) {
        EXPECT(st[i].ppr_sum_fix == FORA_FIX_ONE);       /* mass conserved exactly */
        EXPECT(st[i].n_walks > 0 && st[i].n_idx_hit == 0 && st[i].rsum > 0 && st[i].rsum < 1);
    }
    CHECK(fora_hip_query_batch(ctx, queries, 1, 0, ppr, st)); /* the dense ppr of one query (CHECK_PPR_VALUES) */
    {
        double sum = 0;
        for (int v = 0; v < N; v++) { EXPECT(ppr[v] >= 0); sum += ppr[v]; }
        EXPECT(fabs(sum - 1.0) < 1e-12);
        EXPECT(ppr[queries[0]] >= 0.2 - 1e-12);          /* the source keeps at least alpha */
    }
    fora_timing tm;
    CHECK(fora_hip_get_timing(ctx, &tm));
    EXPECT(tm.batches >= 2 && tm.walks > 0);

    /* ---- build() and hand the index over as deserialize_idx() would */
    uint64_t total = 0;
    static uint64_t off[N], cnt[N];
    CHECK(fora_hip_build_index(ctx));
    CHECK(fora_hip_index_sizes(ctx, &total, off, cnt));
    int32_t *rw_idx = (int32_t *)malloc(sizeof(int32_t) * (total ? total : 1));
    CHECK(fora_hip_get_index(ctx, rw_idx, total, off, cnt));
    CHECK(fora_hip_clear_index(ctx));
    CHECK(fora_hip_set_index(ctx, rw_idx, total, off, cnt));
    CHECK(fora_hip_query_batch(ctx, queries, NQ, /*with_idx*/ 1, NULL, st));
    for (int i = 0; i < NQ; i++) EXPECT(st[i].ppr_sum_fix == FORA_FIX_ONE && st[i].n_idx_hit == st[i].n_walks);
    free(rw_idx);

    /* ---- the topk() loop (both drivers) and gen_exact_topk */
    static int32_t ids[NQ * K], rounds[NQ];
    static double scores[NQ * K];
    CHECK(fora_hip_clear_index(ctx));
    CHECK(fora_hip_topk_bound_batch(ctx, queries, NQ, K, 0.5, 1.0, 0.77, 0, ids, scores, rounds));
    for (int i = 0; i < NQ; i++)
        for (int j = 1; j < K; j++) EXPECT(scores[i * K + j] <= scores[i * K + j - 1]);
    CHECK(fora_hip_set_params(ctx, 0.2, 0.5, 1.0, /*opt*/ 1, 0x464F5241ULL));
    CHECK(fora_hip_topk_batch(ctx, queries, NQ, K, 0.5, 1.0, 0, ids, scores, rounds));
    for (int i = 0; i < NQ; i++) {
        EXPECT(rounds[i] >= 1 && scores[i * K] > 0);
        for (int j = 1; j < K; j++) EXPECT(scores[i * K + j] <= scores[i * K + j - 1]);
    }
    CHECK(fora_hip_power_iteration_batch(ctx, queries, NQ, 100, NULL, NULL, K, ids, scores));
    for (int i = 0; i < NQ; i++) EXPECT(ids[i * K] == queries[i] && scores[i * K] >= 0.2 - 1e-12);

    /* error behaviour: a bad source id is FORA_E_ARG, never a crash (graph.h:155 asserts) */
    int32_t bad = N;
    EXPECT(fora_hip_query_batch(ctx, &bad, 1, 0, NULL, NULL) == FORA_E_ARG);
    fora_hip_destroy(ctx);
    printf("c smoke ok: attach, query, index, top-k and exact PPR through the C ABI\n");
    return 0;
}
