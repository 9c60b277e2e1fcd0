// fora_kernels.h -- device-side data layout and kernels of the FORA SSPPR engine (gfx950, wave64).
//
// Hot path of wangsibovictor/fora re-designed for MI355X (citations into the reference):
//   forward push    algo.h:954-1018   -> k_pushq_bin<NB> + k_accum<false>, one pair per level and bin pass
//                                        (k_push_pop + k_push_expand: the one-atomic-per-edge form, test reference)
//   walk allocation query.h:270-287   -> k_walk_alloc<MODE>
//   random walks    algo.h:124-166, query.h:288-323 -> k_walk_idx<NB>, k_walk_dg (online walks over the degree-grouped copy of the
//                                        graph, narrow layout: one gather per step), k_walk_online<MODE> (wide layouts,
//                                        index build), k_accum<true>
//   index build     build.h:325-354   -> k_index_alloc + k_walk_online<WALK_TO_INDEX>
//   top-k           query.h:972-1045, algo.h:578-610 -> k_topk_frontier, k_count_above, k_topk_select
// Many source queries ("slots") run concurrently; slot q owns dense slabs residue[q*n .. (q+1)*n) and
// ppr[q*n ..) of 2^-62 fixed-point u64 in HBM.  Every cross-thread accumulation is an integer add (LDS
// ds_add_u64 inside a workgroup that owns the target range, integer atomics otherwise), so results do not
// depend on the order adds land in: the whole path is bit-reproducible and checked bit for bit against
// oracle/fora_twin.c.  Global atomics run at ~23 G/s chip-wide on MI355X whatever their type or scope
// (profiles/r01_atomics_microbench.txt), hence the LDS-bucketed organisation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "fora_diag.h"

namespace fora {

constexpr int BLOCK = 256;          // 4 wave64 per workgroup
constexpr uint32_t PUSH_SEG = 256;  // max edges per push work item (bounds hub skew)
constexpr uint32_t WALK_SEG = 1024; // max walks per walk work item
// Schedules that were built, measured and lost (threshold rounds -- options rounds / round_div --, bounded
// deferral -- defer / defer_min) are compiled out of the product library: TEST_PATHS is a compile-time false, the fields
// they read stay in Dev but are never loaded, and the hot kernels do not carry their registers (round 4: 19 / 16 / 10
// SGPR spills in k_push_tail / k_walk_online<0> / k_walk_dg).  -DFORA_TEST_PATHS=1 builds them in: libfora_hip_test.so,
// which the twin-equivalence tests of those schedules load (tests/conftest.py engine_test).
#ifndef FORA_TEST_PATHS
#define FORA_TEST_PATHS 0
#endif
constexpr bool TEST_PATHS = FORA_TEST_PATHS != 0;
constexpr uint64_t FIX_ONE = 1ull << 62;
constexpr uint32_t DEG_SAT = 0xFFFFFFu; // rowinfo low 24 bits: out-degree, saturating
constexpr int MAX_LEVELS = 1 << 15;
// bucketed push (graphs of up to MAX_BINS * BIN_SIZE nodes): increments are binned by target
// range and reduced in LDS instead of one global atomic per edge
#ifndef FORA_BIN_SHIFT
#define FORA_BIN_SHIFT 13
#endif
constexpr int BIN_SHIFT = FORA_BIN_SHIFT;      // narrow layout
constexpr uint32_t BIN_SIZE = 1u << BIN_SHIFT; // 8192 nodes -> 64 KiB of u64 accumulators in LDS
// Wide layouts: 16384-node bins (128 KiB of accumulators, one 1024-thread accumulate workgroup per CU): half the bins, so
// twice the messages per (chunk, bin) run, and a Twitter-2010-sized graph (2543 bins) needs ONE bin pass per level instead
// of two.  Same run: LJ-sized 280 indexed queries 904 -> 833 ms, Twitter-2010-sized 15.96 -> 18.48 q/s.
#ifndef FORA_BIN_SHIFT_WIDE
#define FORA_BIN_SHIFT_WIDE 14
#endif
constexpr int BIN_SHIFT_WIDE = FORA_BIN_SHIFT_WIDE;
constexpr uint32_t BIN_SIZE_WIDE = 1u << BIN_SHIFT_WIDE;
constexpr int MAX_BINS = 128;       // narrow layout: 4-B push messages, staged walk results
constexpr int MAX_BINS_WIDE = 1024; // wide layout: 8-byte messages (local target | value << 14), up to 1024 bins per pass ...
constexpr int MAX_BINS_HUGE = 2560; // ... or 2560 for graphs with more bins (Twitter-2010: 2543 bins, one pass)
#ifndef FORA_ACC_THREADS
#define FORA_ACC_THREADS 512
#endif
constexpr int ACC_THREADS = FORA_ACC_THREADS;
#ifndef FORA_ACC_THREADS_WIDE
#define FORA_ACC_THREADS_WIDE 1024
#endif
constexpr int ACC_THREADS_WIDE = FORA_ACC_THREADS_WIDE; // 16 nodes per lane in the sweep, as in the narrow layout (512 threads: accumulate 214 -> 265 ms on the LJ-sized graph)
constexpr int BIN_EPT = 8; // edges per thread per chunk in k_pushq_bin / k_walk_idx
// wide layouts: edges (index walks) per thread per chunk.  A chunk's messages are written out in one run per bin; with
// hundreds to thousands of bins a run is a few messages -- partial-line writes (Twitter-2010-sized, 8 per thread:
// 3.2 messages per run, WRITE_SIZE 1.77 x the payload) -- and twice the chunk is twice the run.
#ifndef FORA_BIN_EPT_WIDE
#define FORA_BIN_EPT_WIDE 12
#endif
#ifndef FORA_IDX_EPT_WIDE
#define FORA_IDX_EPT_WIDE 12
#endif
// wide layouts: every (chunk, bin) run of messages is padded with null words to a multiple of this many messages, so
// that it starts and ends on a 32-byte sector boundary (4 x 8 bytes).  Writes move whole sectors: an unaligned run of
// R bytes costs (R + 24) / 32 sectors (profiles/r04_pmc_calibration.txt), and aligned sector writes are several times
// faster than unaligned ones.  k_accum skips null words.  1: no padding.
#ifndef FORA_RUN_PAD_WIDE
#define FORA_RUN_PAD_WIDE 1
#endif
constexpr int SEG_BITS = 32 - BIN_SHIFT; // narrow push message = (target & (BIN_SIZE-1)) << SEG_BITS | frontier position
// bucket messages are read exactly once: non-temporal loads keep them from displacing the increment table and the
// slabs in L2 (accumulate kernels -1 %)
#define NT_LOAD(p) __builtin_nontemporal_load(p)
constexpr int CSTRIDE = 32; // u32 words between hot global counters: one 128-B line each
constexpr int SLAB_UNROLL = 4; // slab sweeps: loads in flight per lane
// Tiles of k_pushq_bin / k_walk_idx are made of GRAN-entry granules, one from each of NT / GRAN equal segments
// of the list, instead of NT consecutive entries: neighbouring entries are neighbouring nodes (k_accum writes the
// frontier in node order, k_walk_alloc the items), whose rows / index walks sit side by side in memory, and a tile that
// streams ONE contiguous stretch keeps a few memory channels busy where the granules of distant stretches spread over
// all of them (LJ-sized, 280 indexed queries: k_walk_idx 220 -> 160 ms).
// Granules: the wide bin kernels 64 entries (Twitter-2010-sized 657 -> 627 ms per 28 queries; 16: LJ-sized 359 -> 377 ms),
// the narrow one keeps whole tiles (ws-sized: col is cache-resident, 38.1 ms against 39.6 with 64 and 42.8 with 16),
// k_walk_idx 8 items (Twitter-2010-sized 563 ms whole tiles, 456 with 64, 404 with 16, 387 with 8, 384 with 4).
#ifndef FORA_TILE_GRAN_BIN
#define FORA_TILE_GRAN_BIN 64
#endif
#ifndef FORA_TILE_GRAN_WALK
#define FORA_TILE_GRAN_WALK 8
#endif
// list position of entry `own` (0 .. NT-1) of tile `tile`; seg_len = tiles of the list * GRAN
template <uint32_t GRAN>
__device__ __forceinline__ uint32_t tile_pos(uint32_t own, uint32_t tile, uint32_t seg_len) {
    return (own / GRAN) * seg_len + tile * GRAN + (own % GRAN);
}
#ifndef FORA_BIN_THREADS_WIDE
#define FORA_BIN_THREADS_WIDE 512
#endif
constexpr int BIN_THREADS_WIDE = FORA_BIN_THREADS_WIDE; // wide bin kernel: 512 threads -> 4096-edge chunks, twice the messages per (chunk, bin) run
#ifndef FORA_BIN_THREADS_HUGE
#define FORA_BIN_THREADS_HUGE 1024
#endif
constexpr int BIN_THREADS_HUGE = FORA_BIN_THREADS_HUGE; // more than 1024 bins per pass: 8192-edge chunks (Twitter-2010-sized, 28 queries: bin kernel 777 -> 680 ms)
template <int NB> struct BinThreads { static constexpr int value = NB > 1024 ? BIN_THREADS_HUGE : NB > 128 ? BIN_THREADS_WIDE : 256; };
constexpr int MAX_SUB = 128; // sub-buckets per (slot, bin) bucket = producer workgroups per slot
// Narrow layout: a walk result travels as ONE 64-bit word, node id (< 2^20) | weight << 20 (weights are r / num_s_rw,
// about 2^62 / omega; the rare weight of 2^44 or more goes by a direct atomic).
constexpr int WPACK_SHIFT = 20;
constexpr uint64_t WPACK_MAXW = 1ull << (64 - WPACK_SHIFT);
// Wide layout: every message (push increment or walk result) is ONE 64-bit word, local target (BIN_SHIFT_WIDE bits) | value <<
// BIN_SHIFT_WIDE.  A value of 2^50 or more (the increments of the first one or two levels, about 1e-3 and up) goes through the
// slot's overflow list (push) or a direct atomic (walk weight) and leaves a null word behind.
constexpr uint64_t WIDE_MAXV = 1ull << (64 - BIN_SHIFT_WIDE);

// error flag bits (Dev::err)
constexpr uint32_t ERR_WL_OVERFLOW = 1, ERR_SEG_OVERFLOW = 2, ERR_WIT_OVERFLOW = 4, ERR_BUCKET_OVERFLOW = 8;

struct PushSeg {      // one <=PUSH_SEG-edge slice of a popped node's out-edges
    int64_t ebeg;     // first edge (index into col)
    uint64_t inc;     // increment every target gets: ((1-alpha)*r)/outdeg, algo.h:1002
    uint32_t q;       // slot
    uint32_t cnt;     // edges in this slice
};

struct WalkItem {     // <=WALK_SEG walks that start at one residue node: what the walk kernels work with
    uint64_t j0;      // walk number of the first walk (Philox counter word): a multiple of WALK_SEG
    uint64_t idx_pos; // MODE ppr: rw_idx position of walk j0; MODE index: output position
    uint64_t incr;    // weight per walk: r / num_s_rw (query.h:283-285)
    uint64_t rem;     // walks j < rem carry one extra 2^-62 unit (keeps the sum exact)
    uint32_t v;       // start node
    uint32_t cnt;     // walks in this item (1 .. WALK_SEG)
    uint32_t idx_n;   // leading walks of the item served from the index (query.h:290-307)
};
// ... and how an item lies in memory: 32 bytes (48 until round 6).  With the index nearly every node of an LJ- or Twitter-2010-sized
// slot is an item: 33 GB of items per LJ-sized batch, written by k_walk_alloc and read by k_walk_idx -- a sixth of the latter's traffic.
struct WalkItemP {
    uint64_t w0;      // v (32) | cnt - 1 (10) << 32 | idx_n (11) << 42
    uint64_t w1;      // j0 / WALK_SEG (28) | rem (36) << 28
    uint64_t idx_pos;
    uint64_t incr;
};
static_assert(sizeof(WalkItemP) == 32, "two 16-byte words");
__device__ __forceinline__ WalkItemP wit_pack(const WalkItem &w) {
    WalkItemP p;
    p.w0 = (uint64_t)w.v | ((uint64_t)(w.cnt - 1u) << 32) | ((uint64_t)w.idx_n << 42);
    p.w1 = (w.j0 / WALK_SEG) | (w.rem << 28); // (rem < the node's walk count < 2^36, j0 < 2^38: checked by k_walk_alloc -> ERR_WIT_OVERFLOW)
    p.idx_pos = w.idx_pos;
    p.incr = w.incr;
    return p;
}
__device__ __forceinline__ WalkItem wit_unpack(uint64_t w0, uint64_t w1, uint64_t idx_pos, uint64_t incr) {
    WalkItem w;
    w.v = (uint32_t)w0;
    w.cnt = ((uint32_t)(w0 >> 32) & 1023u) + 1u;
    w.idx_n = (uint32_t)(w0 >> 42) & 2047u;
    w.j0 = (w1 & ((1ull << 28) - 1)) * WALK_SEG;
    w.rem = w1 >> 28;
    w.idx_pos = idx_pos;
    w.incr = incr;
    return w;
}
__device__ __forceinline__ WalkItem wit_load(const WalkItemP *p) {
    const uint4 a = ((const uint4 *)p)[0], b = ((const uint4 *)p)[1];
    return wit_unpack(((uint64_t)a.y << 32) | a.x, ((uint64_t)a.w << 32) | a.z, ((uint64_t)b.y << 32) | b.x, ((uint64_t)b.w << 32) | b.z);
}

struct QState {       // per-slot accumulators
    unsigned long long reserved; // sum of reserve so far; rsum_fix = FIX_ONE - reserved (algo.h:992)
    // dangling mass waiting to return to the source (algo.h:994).  Bucketed push: [L & 1] is delivered by the accumulate
    // of level L while that same launch collects the mass of the nodes it pops for level L + 1 in the other word;
    // the direct path uses [0] only
    unsigned long long dang[2];
    unsigned long long pops, relax;
    unsigned long long n_walks, n_hit, n_rw, ppr_sum;
    uint32_t levels, dangling_source;
    uint32_t tshift, peak;       // bucketed push: the slot's current threshold is t1 << tshift (threshold rounds, see k_round_sweep); largest frontier of the round so far
};

// Degree-grouped copy of the graph for the online walks (k_walk_dg).  Nodes are renumbered so that the out-degree of a
// node follows from its new id alone: ids 0 .. H-1 are the H nodes of largest out-degree, one record each; after them
// every distinct out-degree is a class whose nodes sit side by side (ties in the original order), padded with unused
// ids to a multiple of 2^ts, so that block (id - H) >> ts belongs to ONE class.  A row is deg consecutive entries of the
// bit-packed target list `colp` (targets in copy ids, file order inside a row, algo.h:134-137), rows in copy-id order:
//   first edge of id = base[r] + (id - first[r]) * deg[r],  r = id < H ? id : H + T[(id - H) >> ts]
// with the record tables and T in LDS.  A walk step is then ONE gather (the chosen target) instead of two (row
// offsets, then target), and the row-offset table drops out of the randomly read working set.
struct WalkDG {
    const uint32_t *perm; // [n] original id -> copy id
    const uint32_t *inv;  // [np] copy id -> original id (padding ids: unused)
    const uint32_t *colp; // bit-packed targets, `bits` bits each, read as two aligned dwords; null: no such copy
    const uint32_t *rec;  // [3][nrec]: first copy id | out-degree | first edge, hub records 0 .. H-1, then one per class
    const uint8_t *T;     // [nblk (rounded up to 4)] class of block (id - H) >> ts
    uint32_t H, nrec, nblk, ts, bits;
    uint32_t zero_first;  // first copy id of the out-degree-0 class (np when there is none): ids from here on are dangling
    uint32_t bits32;      // != 0: bit offsets fit 32 bits
    // Results without a gather per walk (k_walk_dg<.., XL = true>): an endpoint among the H hubs -- a quarter to a third of
    // all endpoints -- is added to per-workgroup LDS accumulators (flushed with H atomics per workgroup); the others
    // travel in BUCKET ORDER: the 64-id blocks behind the hubs are dealt round-robin to nbx bins of BIN_SIZE entries
    // (copy ids descend by degree, whole ranges of them are hot: contiguous bins would be badly skewed), word =
    // bin << BIN_SHIFT | local, and k_accum<true> finds the original id of (bin, local) in `invb`, read front to back.
    const uint32_t *invb; // [nbx * BIN_SIZE] bucket order -> original id (unused entries: never touched)
    uint32_t nbx, nbx_magic; // bins of the bucket order; floor(2^32 / nbx) + 1 (exact block / nbx for block < 2^16)
};

struct Dev {
    int32_t n;
    int32_t nq; // slots in use this batch
    const uint64_t *rowinfo; // (first edge << 24) | min(outdeg, DEG_SAT): one 8-B load per node
    const int64_t *row_ptr;
    const int32_t *col;
    const uint32_t *deg;
    // compact copy for the walk steps (nnz < 2^31): 32-bit row offsets and ceil(log2 n)-bit packed
    // column entries shrink the randomly gathered working set (ws: 11.2 MB -> 6.6 MB) for a better
    // L2 hit rate -- the walk kernel is bound by L2 misses, each of which moves a 64-B line
    const uint32_t *rp32;   // [n + 1] or null
    const uint32_t *colp;   // bit-packed column ids, entry e at bit e*colbits, read as two aligned dwords
    uint32_t colbits;
    uint32_t colp32;        // != 0: bit offsets fit 32 bits
    WalkDG dg;              // degree-grouped copy (narrow layout, see k_walk_dg)
    const uint32_t *acc_xl; // k_accum<true>: the buckets hold results in WalkDG bucket order, original id = acc_xl[bin << BIN_SHIFT | local]; null: plain ids
    uint64_t *residue, *ppr;
    uint64_t *wl[2];
    uint64_t wl_cap;
    PushSeg *seg;
    uint64_t seg_cap;
    WalkItemP *wit;         // [slot][wit_cap] walk items of a slot
    uint64_t wit_cap;       // per slot
    uint32_t *wit_count;    // [slot * CSTRIDE]
    unsigned long long *wl_count;  // [MAX_LEVELS + 2] frontier size per level
    unsigned long long *seg_count; // [MAX_LEVELS + 2]
    unsigned long long *tot_steps; // [1]
    QState *qs;
    const int32_t *src; // source node per slot
    uint32_t *err;
    uint64_t afix; // alpha * 2^62
    uint64_t t1;   // ceil(rmax * 2^62)
    uint32_t alpha32; // floor(alpha * 2^32): Bernoulli(alpha) stop threshold
    uint32_t seed_lo, seed_hi;
    double alpha, omega;
    int32_t opt;
    const int32_t *rw_idx;
    const uint64_t *idx_off, *idx_cnt;
    // ---- bucketed push state (binned != 0)
    int32_t binned, nbins;  // nbins: bins of the whole graph
    int32_t pbins;          // bins per pass = stride of the bucket arrays (<= MAX_BINS_WIDE)
    int32_t bin_lo, bin_cnt; // bins [bin_lo, bin_lo + bin_cnt) are handled by the current pass
    // multi-pass graphs: a copy of col with every row sorted by target (the push sums over a row, order is
    // free) and, per node, the offsets where each pass's target range starts -- a pass then reads only its own
    // part of every popped row instead of the whole row
    const int32_t *col_push;   // == col when there is one pass
    const uint32_t *row_split; // [n][npass + 1] or null
    int32_t npass, pass;
    // Hub pre-aggregation (single-pass layouts).  The `hubs` nodes of largest in-degree receive a large share of all
    // increments (ws-sized R-MAT: the top 4096 of 282 k nodes are the target of 45 % of the edges).  In a level whose
    // frontier holds at least hub_min nodes the bin kernel reads `col_hub` -- col with hub targets replaced by
    // 0x80000000 | hub index -- and adds an increment for a hub to a per-workgroup LDS accumulator instead of emitting a
    // message; the sums leave the workgroup as one dense row hubsum[slot][workgroup][hub] (plain coalesced stores), and
    // the accumulate of the hub's bin adds the `sub` partial sums to its LDS accumulators like any other message.  Hub
    // indices follow the node ids, so the hubs of one bin are a contiguous range [hub_first[b], hub_first[b + 1]).
    // Integer adds commute: same bits as the plain messages.
    const int32_t *col_hub;    // [nnz] or null
    // Quad copies for the wide bin kernel (round 5): col / col_hub with every row padded with -1 words to whole quads (16-byte
    // aligned) and rowinfo4 = (first QUAD of the row << 24) | min(outdeg, DEG_SAT): a lane of k_pushq_bin<.., QUAD> reads four
    // consecutive edges with ONE 16-byte load (a vector memory instruction costs the CU's address path the same whatever it
    // carries).  Null: the kernel reads single edges (narrow layout, several bin passes per level).
    const int32_t *col4, *col_hub4;
    const uint64_t *rowinfo4;
    const uint32_t *hub_node;  // [hubs] node id of hub h
    const uint32_t *hub_first; // [nbins + 1]
    uint64_t *hubsum;          // [slot][sub][hubs]
    uint32_t hubs, hub_min;
    uint32_t tail_hubs;     // != 0: k_push_tail sums the increments for hubs in LDS too (its launches pass hubs * 8 bytes of dynamic LDS)
    uint32_t tiny_max; // k_accum: buckets of up to this many messages go by direct atomics instead of the LDS sweep
    uint32_t slot_major; // wide layouts, one bit per kernel (1: k_pushq_bin, 2: k_accum, 4: k_walk_idx, 8: k_walk_alloc): the launch puts the SLOT in blockIdx.x
                         // (the fastest-varying index of the dispatch order) and the tile / bin / chunk in blockIdx.y, so that the workgroups in flight at one time
                         // belong to many slots instead of a few.  Measured, not derived (round 6, profiles/r06_slot_major.txt): LJ-sized launches of 143 slots gain
                         // 4-9 % in the bin kernel and ~5 % in the indexed walks; HBM-side bytes and L2 hit counts are the same in both orders, the
                         // requests are served faster (Infinity Cache reuse is the hypothesis that fits)
    uint32_t acc_group; // k_accum (wide layouts): consecutive bins of a slot per workgroup (>= 1).  The workgroup reads their counts in one coalesced trip and
                        // skips the bins that have nothing to do: 94 k one-bin workgroups per launch cost 0.74 ms when nearly all of them are empty (sparse levels,
                        // top-k rounds)
    uint32_t *fl[2];        // [slot][n] frontier node lists, ping-pong by level parity
    uint32_t *fl_count[2];  // [slot]
    // [slot][segq_cap] increment of the node at frontier position i, ping-pong like fl: the accumulate of level L
    // gathers from [L & 1] while it writes the entries of the nodes it pops for level L + 1 into the other one
    uint64_t *inc_tab[2];
    unsigned long long *stamps; // diagnostic builds (-DFORA_STAMPS): cycles per kernel phase, [0..15] bin kernel, [16..31] accumulate
    int32_t rounds;         // threshold rounds of the bucketed push (k_round_sweep); 1: the plain schedule
    uint32_t round_div;     // a round is left once its frontier is down to 1/round_div of its largest one (0: only when empty)
    // Bounded deferral (option "defer", oracle/fora_twin.c twin_levels_div): a node that crosses its threshold in level
    // L but ends the level with less than 2^defer_k times the threshold is not popped in level L + 1: it stays where it
    // is, is marked in the slot's bitmap, and the accumulate of level L + 1 -- which owns its residue word anyway -- hands it
    // to the frontier of level L + 2 with whatever it holds then.  Bitmaps ping-pong by level parity: the accumulate of
    // level L reads (and clears) [L & 1], writes [(L + 1) & 1]; a word covers 64 consecutive nodes of one bin, so it has
    // exactly one writer (the wave that sweeps those nodes).  dflag: the bin has marks at all (early exit of k_accum).
    // Word 2 of a slot's fl_count line counts the marks for the host's termination test.  k_push_tail keeps lists (dl).
    int32_t defer_k;        // 0: plain levels
    uint32_t defer_min;     // only the levels that pop at least this many nodes of the slot defer (the long tail of small levels gains nothing from it)
    uint64_t *dbm[2];       // [slot][dbm_words]
    uint32_t dbm_words;     // per slot: nbins << (bin shift - 6)
    uint32_t *dflag[2];     // [slot][nbins]
    uint32_t *dl[2];        // [slot][n] k_push_tail: nodes deferred by the previous / this level
    uint32_t *sw_count, *sw_done; // [slot * CSTRIDE] k_round_sweep: entries appended / workgroups finished (both return to 0)
    // wide bin kernels: tiles beyond a workgroup's first are dealt out through a per-slot counter, so that the workgroup
    // that meets a hub's row (Twitter-2010-sized: up to 1.4 M edges, 170 chunks) simply takes fewer other tiles -- with
    // 7 slots x 128 workgroups there is barely more than one round of workgroups per launch to even things out.
    // Ping-pong by launch parity: a launch zeroes the counter the next one will use.
    uint32_t *tile_ctr[2];        // [slot * CSTRIDE]
    int32_t launch_par;
    int32_t pop_next;       // k_accum<false>: pop the crossing nodes for the next level (0 on the last level of a capped run)
    uint64_t segq_cap;      // = n: a frontier holds each node at most once
    // Message buckets.  Every (slot, bin) bucket is cut into `sub` sub-buckets, one per producer workgroup of the slot
    // (the bin kernel and the walk kernels launch exactly `sub` workgroups per slot): workgroup x appends only to
    // sub-bucket x and keeps its fill counters in LDS, so an append costs no global atomic at all -- with hundreds of
    // bins a 2048-edge chunk holds 2-4 messages per bin, and one reservation atomic per (chunk, bin) ran the wide bin
    // kernel into the chip's ~23 G/s memory-side atomic rate.  The counters are loaded when a producer kernel starts
    // and stored when it ends; k_accum reads the `sub` counts of its bucket and zeroes them.
    uint32_t *bk_w;         // [slot][bin][sub][bk_cap] target node of a pending increment
    uint64_t *bk_inc;       // [slot][bin][sub][bk_cap] its value
    uint32_t *bk_count;     // [slot][bin][sub]
    uint32_t bk_cap;        // capacity of ONE sub-bucket
    uint32_t sub;           // sub-buckets per bucket (<= MAX_SUB)
    // messages that found their bucket full (rare; capacity is a tuning knob): per-slot overflow list,
    // folded in by k_accum.  Count is double-buffered by level parity (zeroed one level later).
    uint32_t *ov_w;         // [slot][ov_cap]
    uint64_t *ov_inc;       // [slot][ov_cap]
    uint32_t *ov_count[2];  // [slot * CSTRIDE]
    uint32_t *ov_bin[2];    // [slot][nbins] entries of the list that belong to the bin: k_accum scans the list only when its own count is not zero
    uint32_t ov_cap;
    int32_t wide;           // != 0: nbins > MAX_BINS, push messages are (bk_w = local target, bk_inc = increment)
};

// ------------------------------------------------------------------ helpers
// Diagnostic build: thread 0 of a workgroup adds the shader-clock cycles since the previous stamp to d.stamps[slot].
// (phase stamps and probes: fora_diag.h -- empty in the product build)
__device__ __forceinline__ uint64_t mulshift62(uint64_t r, uint64_t a) {
    return (__umul64hi(r, a) << 2) | ((r * a) >> 62);
}
__device__ __forceinline__ double fix2d(uint64_t x) { return (double)x * 0x1p-62; }

// exact floor(a / b) for a < 2^63, 1 <= b < 2^32 without the ~150-instruction software 64-bit divide:
// f64 reciprocal estimate (|error| <= 1536/b + 1), one exact remainder step, +-1 fix-up.
__device__ __forceinline__ uint64_t div_u64_u32(uint64_t a, uint32_t b) {
    const double rb = 1.0 / (double)b;
    const uint64_t q1 = (uint64_t)((double)a * rb);
    const int64_t r1 = (int64_t)(a - q1 * (uint64_t)b); // |r1| < 2^12 * b: exact in f64
    int64_t q2 = (int64_t)floor((double)r1 * rb);
    int64_t r2 = r1 - q2 * (int64_t)b;
    if (r2 < 0) { q2--; r2 += b; }
    if (r2 >= (int64_t)b) { q2++; }
    return q1 + (uint64_t)q2;
}

// integer form of "residue/outdeg >= rmax" (algo.h:1012); outdeg 0 -> any residue > 0
__device__ __forceinline__ uint64_t node_thr(uint64_t t1, uint32_t deg) {
    if (deg == 0) return 1;
    uint64_t hi = __umul64hi(t1, (uint64_t)deg);
    return hi ? ~0ull : t1 * (uint64_t)deg;
}

// One pop (algo.h:983-1002): the node gives up residue r, keeps alpha of it as reserve, and every out-edge gets
// inc = ((1-alpha)*r)/outdeg (returned).  The integer-division remainder stays reserved; a dangling node's
// (1-alpha)*r goes back to the source (algo.h:993-994) through `dang`.
__device__ __forceinline__ uint64_t pop_value(uint64_t afix, uint64_t r, uint32_t deg, uint64_t &res_add, uint64_t &dang) {
    const uint64_t keep = mulshift62(r, afix); // v_residue * alpha
    const uint64_t push = r - keep;            // (1-alpha) * v_residue
    if (deg == 0) { res_add = keep; dang = push; return 0; }
    const uint64_t inc = div_u64_u32(push, deg);
    res_add = keep + (push - inc * deg);
    dang = 0;
    return inc;
}

// threshold unit of a round: t1 << k, saturating
__device__ __forceinline__ uint64_t thr_unit(uint64_t t1, uint32_t k) {
    return k == 0 ? t1 : ((t1 >> (63 - k)) ? (~0ull >> 1) : (t1 << k));
}

__device__ __forceinline__ void node_row(const Dev &d, uint32_t v, int64_t &beg, uint64_t &deg) {
    uint64_t ri = d.rowinfo[v];
    beg = (int64_t)(ri >> 24);
    deg = ri & DEG_SAT;
    if (deg == DEG_SAT) deg = (uint64_t)(d.row_ptr[v + 1] - beg);
}

// exact out-degree from a rowinfo word (the 24-bit field saturates on hubs)
__device__ __forceinline__ uint32_t ri_deg(const Dev &d, uint64_t ri, uint32_t v) {
    const uint32_t dg = (uint32_t)ri & DEG_SAT;
    return dg == DEG_SAT ? (uint32_t)(d.row_ptr[v + 1] - (int64_t)(ri >> 24)) : dg;
}

struct __attribute__((packed, aligned(4))) U32Pair { uint32_t a, b; }; // dword-aligned pair: one dwordx2 load
// column id of edge e from the bit-packed copy
__device__ __forceinline__ uint32_t colp_at(const Dev &d, uint64_t e) {
    uint32_t word, sh;
    if (d.colp32) { // nnz * colbits < 2^32: one 32-bit multiply instead of a 64-bit one (quarter-rate on the VALU)
        const uint32_t at = d.colbits * (uint32_t)e;
        word = at >> 5; sh = at & 31;
    } else {
        const uint64_t at = (uint64_t)d.colbits * e;
        word = (uint32_t)(at >> 5); sh = (uint32_t)at & 31;
    }
    const U32Pair cw = *(const U32Pair *)(d.colp + word);
    const uint64_t both = ((uint64_t)cw.b << 32) | cw.a;
    return (uint32_t)(both >> sh) & ((1u << d.colbits) - 1u);
}
// one walk move from `cur` with random word wm: returns the chosen out-neighbour, or `start` when
// cur is dangling (algo.h:134-140)
__device__ __forceinline__ uint32_t walk_move(const Dev &d, uint32_t cur, uint32_t start, uint32_t wm) {
    if (d.rp32) {
        const U32Pair rp = *(const U32Pair *)(d.rp32 + cur);
        const uint32_t dg = rp.b - rp.a;
        if (!dg) return start;
        return colp_at(d, (uint64_t)rp.a + __umulhi(wm, dg));
    }
    int64_t beg; uint64_t deg;
    node_row(d, cur, beg, deg);
    if (!deg) return start;
    return (uint32_t)d.col[beg + (int64_t)(((uint64_t)wm * deg) >> 32)];
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// Cross-lane moves on the VALU (DPP, gfx9 controls): row_shr:n = 0x110 + n shifts inside a row of 16 lanes, row_bcast:15 /
// row_bcast:31 (0x142 / 0x143) hand the last lane of a row / of the lower half to the lanes above, wave_shr:1 (0x138) shifts
// the whole wave by one.  Lanes without a source (and rows outside ROWS) read 0.  A wave scan is six of these instead of six
// LDS round trips (ds_bpermute).
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xF, false); }
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t x) {
    x += dpp0<0x111>(x); x += dpp0<0x112>(x); x += dpp0<0x114>(x); x += dpp0<0x118>(x);
    x += dpp0<0x142, 0xA>(x);
    x += dpp0<0x143, 0xC>(x);
    return x;
}
__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t x) {
    x = max(x, dpp0<0x111>(x)); x = max(x, dpp0<0x112>(x)); x = max(x, dpp0<0x114>(x)); x = max(x, dpp0<0x118>(x));
    x = max(x, dpp0<0x142, 0xA>(x));
    x = max(x, dpp0<0x143, 0xC>(x));
    return x;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
    const uint32_t x = wave_incl_scan_add(v);
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}
// exclusive scan over the NT threads of a block; s_w: NT / 64-entry LDS scratch
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_n(uint32_t v, uint32_t *s_w, uint32_t &total) {
    uint32_t wtot;
    uint32_t x = wave_excl_scan(v, wtot);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[w] = wtot;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) { const uint32_t c = s_w[i]; if (i < w) before += c; tot += c; }
    total = tot;
    return x + before;
}
// exclusive scan over the 256 threads of a block; s_w: 4-entry LDS scratch
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t &total) {
    uint32_t wtot;
    uint32_t x = wave_excl_scan(v, wtot);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[w] = wtot;
    __syncthreads();
    uint32_t a0 = s_w[0], a1 = s_w[1], a2 = s_w[2], a3 = s_w[3];
    total = a0 + a1 + a2 + a3;
    return x + (w > 0 ? a0 : 0) + (w > 1 ? a1 : 0) + (w > 2 ? a2 : 0);
}

// wave-aggregated append of `flag`ged 64-bit items to a list (one atomic per wave)
__device__ __forceinline__ void wave_append(bool flag, uint64_t item, uint64_t *list,
                                            unsigned long long *count, uint64_t cap, uint32_t *err,
                                            uint32_t errbit) {
    const unsigned long long mask = __ballot(flag);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned long long)__popcll(mask));
    base = __shfl(base, leader);
    if (flag) {
        uint64_t idx = base + __popcll(mask & ((1ull << lane) - 1));
        if (idx < cap) list[idx] = item;
        else atomicOr(err, errbit);
    }
}

// same for 32-bit items and a 32-bit counter (per-slot frontier lists)
__device__ __forceinline__ void wave_append32(bool flag, uint32_t item, uint32_t *list, uint32_t *count,
                                              uint32_t cap, uint32_t *err, uint32_t errbit) {
    const unsigned long long mask = __ballot(flag);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (flag) {
        const uint32_t idx = base + __popcll(mask & ((1ull << lane) - 1));
        if (idx < cap) list[idx] = item;
        else atomicOr(err, errbit);
    }
}

// Philox4x32-10 (Random123 constants)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 product each (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: the walk kernels are
        // bound by these quarter-rate multiplies (profiles/r04_walk_bound.txt)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// walk allocation in the reference's f64 arithmetic: query.h:282 / :314
//   num_s_rw = ceil(residual / check_rsum * num_random_walk)
__device__ __forceinline__ uint64_t walk_count(double residual, double check_rsum, uint64_t N) {
    return (uint64_t)ceil(residual / check_rsum * (double)N);
}

// ------------------------------------------------------------------ init
// residue[s] = 1, frontier = {s} (algo.h:969-978); dangling source: reserve[s] = 1 (algo.h:961-965)
// mode 0: query, 1: top-k (round frontier built by k_topk_frontier), 2: power iteration (no dangling-source
// short cut: query.h:1192-1224 iterates it like any other node), 3: query through k_push_team
__global__ void __launch_bounds__(BLOCK) k_init_batch(Dev d, int mode) {
    int q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= d.nq) return;
    QState z = {};
    const uint32_t s = (uint32_t)d.src[q];
    const uint32_t deg = d.deg[s];
    const uint64_t a = (uint64_t)q * d.n + s;
    if (deg == 0 && mode != 2) {
        d.ppr[a] = FIX_ONE;
        z.reserved = FIX_ONE;
        z.dangling_source = 1;
    } else if (mode == 1) {
        d.residue[a] = FIX_ONE;
    } else if (mode == 3) {
        // team push (fora_team.h): the source's owner sets its residue in LDS
    } else if (d.binned) {
        // bucketed push: a frontier entry is (node, residue taken from it); k_pushq_bin of level 0 pops it
        d.fl[0][(uint64_t)q * d.n] = s;
        d.inc_tab[0][(uint64_t)q * d.segq_cap] = FIX_ONE;
        d.fl_count[0][q * CSTRIDE] = 1;
        if (TEST_PATHS && mode == 0 && d.rounds > 1) { // threshold rounds: start at t1 << (rounds - 1); the host reads the shift beside the frontier size
            z.tshift = (uint32_t)d.rounds - 1;
            d.fl_count[0][q * CSTRIDE + 1] = z.tshift;
            d.fl_count[1][q * CSTRIDE + 1] = z.tshift;
        }
    } else {
        d.residue[a] = FIX_ONE;
        unsigned long long i = atomicAdd(&d.wl_count[0], 1ull);
        d.wl[0][i] = ((uint64_t)q << 32) | s;
    }
    d.qs[q] = z;
}

// ------------------------------------------------------------------ push: pop
// One thread per frontier entry (q, v): take the residue, keep alpha of it, cut the
// out-edges into <=PUSH_SEG slices for k_push_expand (algo.h:983-1002).
__global__ void __launch_bounds__(BLOCK) k_push_pop(Dev d, int L) {
    const uint64_t count = d.wl_count[L];
    const uint64_t *in = d.wl[L & 1];
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < count; base += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t i = base + threadIdx.x;
        const bool act = i < count;
        uint32_t q = 0, nseg = 0;
        uint64_t res_add = 0, dang = 0, inc = 0, deg = 0;
        int64_t beg = 0;
        if (act) {
            const uint64_t item = in[i];
            q = (uint32_t)(item >> 32);
            const uint32_t v = (uint32_t)item;
            const uint64_t a = (uint64_t)q * d.n + v;
            const uint64_t r = d.residue[a];
            d.residue[a] = 0;                                 // algo.h:985
            const uint64_t keep = mulshift62(r, d.afix);      // v_residue * alpha
            const uint64_t push = r - keep;                   // (1-alpha) * v_residue
            node_row(d, v, beg, deg);
            if (deg == 0) {                                   // algo.h:993-994
                res_add = keep;
                dang = push;
            } else {
                inc = deg < (1ull << 32) ? div_u64_u32(push, (uint32_t)deg) : push / deg; // algo.h:1002
                res_add = keep + (push - inc * deg);          // division remainder stays reserved
                nseg = (uint32_t)((deg + PUSH_SEG - 1) / PUSH_SEG);
            }
            d.ppr[a] += res_add;                              // algo.h:986-989 (only this thread owns (q,v))
        }
        // slices -> seg list
        uint32_t tot;
        const uint32_t off = wave_excl_scan(nseg, tot);
        if (tot) {
            unsigned long long sb = 0;
            if (lane == 0) sb = atomicAdd(&d.seg_count[L], (unsigned long long)tot);
            sb = __shfl(sb, 0);
            if (sb + tot > d.seg_cap) {
                if (lane == 0) atomicOr(d.err, ERR_SEG_OVERFLOW);
            } else {
                for (uint32_t k = 0; k < nseg; k++) {
                    PushSeg s;
                    s.ebeg = beg + (int64_t)k * PUSH_SEG;
                    s.inc = inc;
                    s.q = q;
                    const uint64_t left = deg - (uint64_t)k * PUSH_SEG;
                    s.cnt = left < PUSH_SEG ? (uint32_t)left : PUSH_SEG;
                    d.seg[sb + off + k] = s;
                }
            }
        }
        // per-slot accumulators: rsum bookkeeping (algo.h:992), dangling mass, counters
        const uint32_t q0 = __builtin_amdgcn_readfirstlane(q);
        if (__all(!act || q == q0)) {
            const uint64_t sr = wave_sum(res_add), sd = wave_sum(dang);
            const uint64_t sp = wave_sum(act ? 1 : 0), se = wave_sum(deg);
            if (lane == 0 && sp) {
                QState *s = &d.qs[q0];
                atomicAdd(&s->reserved, (unsigned long long)sr);
                if (sd) atomicAdd(&s->dang[0], (unsigned long long)sd);
                atomicAdd(&s->pops, (unsigned long long)sp);
                if (se) atomicAdd(&s->relax, (unsigned long long)se);
                s->levels = (uint32_t)L + 1;
            }
        } else if (act) {
            QState *s = &d.qs[q];
            atomicAdd(&s->reserved, (unsigned long long)res_add);
            if (dang) atomicAdd(&s->dang[0], (unsigned long long)dang);
            atomicAdd(&s->pops, 1ull);
            if (deg) atomicAdd(&s->relax, (unsigned long long)deg);
            s->levels = (uint32_t)L + 1;
        }
    }
}

// ------------------------------------------------------------------ push: expand
// Direct (one atomic per edge) form, FORA_HIP_DIRECT=1 only.  A block stages 256 edge slices in LDS (first edge, increment,
// slot, prefix sum of slice lengths), then its threads walk the concatenated edge
// range: consecutive threads read consecutive col entries (coalesced), add the
// increment to residue[q][w] with one returning u64 atomic, and the thread whose
// add carries the residue across w's threshold appends (q, w) to the next frontier
// (algo.h:1003-1016).  Increments are positive, so exactly one add crosses.
__global__ void __launch_bounds__(BLOCK) k_push_expand(Dev d, int L) {
    __shared__ int64_t s_ebeg[BLOCK];
    __shared__ uint64_t s_inc[BLOCK];
    __shared__ uint32_t s_q[BLOCK];
    __shared__ uint32_t s_pref[BLOCK + 1];
    __shared__ uint32_t s_w[4];
    uint64_t *out = d.wl[(L + 1) & 1];
    unsigned long long *out_count = &d.wl_count[L + 1];

    // dangling mass collected by k_push_pop returns to the source (algo.h:994-998)
    if ((uint64_t)blockIdx.x * BLOCK < (uint64_t)d.nq) {
        const int q = blockIdx.x * BLOCK + threadIdx.x;
        bool cross = false;
        uint64_t item = 0;
        if (q < d.nq) {
            const uint64_t dm = d.qs[q].dang[0];
            if (dm) {
                d.qs[q].dang[0] = 0;
                const uint32_t s = (uint32_t)d.src[q];
                const uint64_t old = atomicAdd((unsigned long long *)&d.residue[(uint64_t)q * d.n + s],
                                               (unsigned long long)dm);
                const uint64_t thr = node_thr(d.t1, d.deg[s]);
                cross = old < thr && old + dm >= thr;
                item = ((uint64_t)q << 32) | s;
            }
        }
        wave_append(cross, item, out, out_count, d.wl_cap, d.err, ERR_WL_OVERFLOW);
    }

    const uint64_t count = d.seg_count[L];
    for (uint64_t tbase = (uint64_t)blockIdx.x * BLOCK; tbase < count; tbase += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t i = tbase + threadIdx.x;
        uint32_t cnt = 0;
        if (i < count) {
            const PushSeg s = d.seg[i];
            s_ebeg[threadIdx.x] = s.ebeg;
            s_inc[threadIdx.x] = s.inc;
            s_q[threadIdx.x] = s.q;
            cnt = s.cnt;
        }
        uint32_t total;
        const uint32_t pre = block_excl_scan(cnt, s_w, total);
        s_pref[threadIdx.x] = pre;
        if (threadIdx.x == 0) s_pref[BLOCK] = total;
        __syncthreads();
        for (uint32_t eb = 0; eb < total; eb += BLOCK) {
            const uint32_t e = eb + threadIdx.x;
            bool cross = false;
            uint64_t item = 0;
            if (e < total) {
                uint32_t lo = 0, hi = BLOCK; // largest lo with s_pref[lo] <= e
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_pref[mid] <= e) lo = mid; else hi = mid;
                }
                const uint32_t w = (uint32_t)d.col[s_ebeg[lo] + (e - s_pref[lo])];
                const uint64_t inc = s_inc[lo];
                const uint32_t q = s_q[lo];
                const uint64_t old = atomicAdd((unsigned long long *)&d.residue[(uint64_t)q * d.n + w],
                                               (unsigned long long)inc);
                const uint64_t thr = node_thr(d.t1, d.deg[w]);
                cross = old < thr && old + inc >= thr;
                item = ((uint64_t)q << 32) | w;
            }
            wave_append(cross, item, out, out_count, d.wl_cap, d.err, ERR_WL_OVERFLOW);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ bucketed push
// Same level semantics as k_push_pop / k_push_expand, reorganised so that no per-edge global
// atomic is needed (measured: every global atomic flavour caps at ~23 G/s chip-wide, ~5 G/s
// with power-law hot targets).  A frontier entry is (node, residue): whoever put it there TOOK the
// residue (the slab word is already 0).  Per level (and per pass of <= 1024 bins on huge graphs):
//   k_pushq_bin     frontier (per-slot list, node-ordered inside every bin's run) -> the pop arithmetic
//                   (alpha of the residue to the reserve slab, increment per out-edge) one entry per
//                   lane, then increments binned by target range: per 2048-edge chunk an LDS
//                   histogram, space in the workgroup's own sub-bucket of every bin (a counter in LDS, no
//                   global atomic), then bin-sorted message stores
//   k_accum<false>  one workgroup per (slot, bin): ds_add_u64 every message into 64 KiB of LDS
//                   accumulators, then sweep them: one plain RMW per touched node (the
//                   workgroup owns that residue range); a node that crosses its threshold has
//                   its residue line in hand: it is stored as 0 and (node, residue) goes to the
//                   next frontier at a rank that follows the NODE ORDER, so the next level reads
//                   rows, reserves and residues almost sequentially and never gathers a residue.
// Integer adds commute, so the result is bit-identical to the direct path and to the twin.

// grid = (X, nq).  A block takes 256 frontier entries (node, residue) of a slot and pops them (residue -> reserve, increment;
// algo.h:983-1002; row start, increment, degree prefix sum into LDS).  Then it bins their concatenated out-edges in
// chunks of BIN_EPT * BLOCK: each lane gathers BIN_EPT consecutive edges (all loads issued before any is waited
// for), an LDS histogram over the target bins gives every message its rank, the workgroup's LDS fill counters give the
// chunk its place in the workgroup's own sub-buckets (Dev::bk_w: no global atomic), the messages are staged bin-sorted
// in LDS and the stage is written out in runs.  No slice list is materialised: a narrow message names the frontier position of its source node
// (increment table `inc_tab`), a wide one carries the increment.  Measured and dropped: gathering with
// consecutive lanes on consecutive edges through an LDS address table, carrying rowinfo in the frontier entry, and
// loading the entries one or two tiles ahead -- the kernel is bound by its instruction and LDS mix, not by these waits.
// HUB: the graph has a hub copy (Dev::col_hub); SPLIT: several bin passes per level over row-split offsets; ROUNDS: threshold
// rounds / bounded deferral bookkeeping.  The plain instantiation does not carry what it does not use (spills).
template <int NB, bool HUB, bool SPLIT, bool SCHED, bool QUAD = false>
__global__ void __launch_bounds__(BinThreads<NB>::value) k_pushq_bin(Dev d, int L) {
    static_assert(!(QUAD && SPLIT), "pass-split rows are read edge by edge");
    constexpr int NT = BinThreads<NB>::value; // workgroup size = frontier entries per tile; BIN_EPT * NT edges per chunk
    const bool slot_major = NB > MAX_BINS && (d.slot_major & 1u); // (see Dev::slot_major)
    const int q = slot_major ? blockIdx.x : blockIdx.y;
    const uint32_t bx = slot_major ? blockIdx.y : blockIdx.x, gx = slot_major ? gridDim.y : gridDim.x; // this workgroup's number among the slot's, and their count
    const int par = L & 1;
    const uint32_t count = d.fl_count[par][q * CSTRIDE];
    const bool first_pass = d.bin_lo == 0; // graphs with more than pbins bins run several bin/accum passes per level
    if (first_pass && bx == 0 && threadIdx.x == 0) {
        d.fl_count[par ^ 1][q * CSTRIDE] = 0; // next level's list starts empty
        if (SCHED) d.fl_count[par ^ 1][q * CSTRIDE + 2] = 0; // ... and so does the count of nodes this level defers
        d.ov_count[par ^ 1][q * CSTRIDE] = 0; // consumed by k_accum of the previous level
        if (count) d.qs[q].levels++;          // levels in which the slot popped (this thread is the only writer in a launch)
        if (SCHED && d.rounds > 1 && count > d.qs[q].peak) d.qs[q].peak = count; // k_round_sweep compares the next frontier with it
    }
    constexpr bool WIDE = NB > MAX_BINS;
    if (WIDE && bx == 0 && threadIdx.x == 0) d.tile_ctr[d.launch_par ^ 1][q * CSTRIDE] = 0; // for the next launch
    if (!count) return;
    // hub pre-aggregation (see Dev::col_hub): the same predicate in k_accum decides whether hubsum is read
    extern __shared__ unsigned long long s_hub[]; // [d.hubs] when hubmode
    const bool hubmode = HUB && d.col_hub && count >= d.hub_min;
    if (hubmode) for (uint32_t i = threadIdx.x; i < d.hubs; i += BinThreads<NB>::value) s_hub[i] = 0;
    const int32_t *colsrc = QUAD ? (hubmode ? d.col_hub4 : d.col4) : (hubmode ? d.col_hub : d.col_push);
    constexpr int BS = WIDE ? BIN_SHIFT_WIDE : BIN_SHIFT; // bits of a local target: 13 narrow, 14 in the wide layouts
    constexpr uint32_t BSZ = 1u << BS;
    constexpr int SRC_BITS = NT == 256 ? 8 : NT == 512 ? 9 : 10;
    static_assert(BS + SRC_BITS <= 32, "wide stage word: local target | source entry");
    constexpr int BIN_EPT = NB > MAX_BINS ? FORA_BIN_EPT_WIDE : fora::BIN_EPT; // (shadows the namespace constant inside this kernel)
    constexpr uint32_t CHUNK = NT * BIN_EPT;
    constexpr uint32_t PAD = NB > MAX_BINS ? FORA_RUN_PAD_WIDE : 1;
    __shared__ int64_t s_ebeg[NT];
    __shared__ uint64_t s_inc[NT];
    __shared__ uint32_t s_pref[NT + 1];
    __shared__ uint32_t s_w[NT / 64];
    // per bin: messages of the chunk, first stage slot of the bin (s_lofs[b + 1] - s_lofs[b] = that count again), and
    // s_fill: messages this workgroup has put into its sub-bucket of the bin so far
    __shared__ uint32_t s_cnt[NB], s_lofs[NB + 1];
    __shared__ uint32_t s_fill[NB];
    BIN_RANK_PROBE_DECL(NB)
    // stage: ONE word per message and its bin.  narrow: (local target << SEG_BITS) | frontier position;
    // wide: local target (13 bits) | source entry inside the tile (SRC_BITS)
    __shared__ uint32_t s_msg[CHUNK];
    __shared__ typename std::conditional<WIDE, uint16_t, uint8_t>::type s_bin[CHUNK];
    const int lane = threadIdx.x & 63;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint64_t fbase = (uint64_t)q * d.segq_cap;
    const uint32_t *in = d.fl[par] + slab;
    uint64_t *incs = d.inc_tab[par] + fbase;
    const uint32_t sub = d.sub; // == gx: this workgroup owns sub-bucket bx of every bin of the slot
    uint32_t *bkc = d.bk_count + (uint64_t)q * d.pbins * sub + bx;                  // count of bin b: bkc[b * sub]
    const uint64_t bk0 = ((uint64_t)q * d.pbins * sub + bx) * d.bk_cap;             // sub-bucket of bin b: bk0 + b * sub * bk_cap
    const uint64_t bstride = (uint64_t)sub * d.bk_cap;
    const uint32_t bin_lo = (uint32_t)d.bin_lo, bin_cnt = (uint32_t)d.bin_cnt;
    for (uint32_t i = threadIdx.x; i < (uint32_t)NB; i += NT) {
        s_cnt[i] = 0;
        s_fill[i] = i < bin_cnt ? bkc[(uint64_t)i * sub] : 0;
    }
    uint64_t acc_res = 0, acc_dang = 0, acc_pops = 0, acc_relax = 0;
    STAMP_DECL
    constexpr uint32_t GRAN = WIDE ? FORA_TILE_GRAN_BIN : NT; // see tile_pos
    const uint32_t ntiles = (count + NT - 1) / NT;
    const uint32_t seg_len = ntiles * GRAN;
    __shared__ uint32_t s_next;
    for (uint32_t tile = bx; tile < ntiles;) {
        uint32_t next_tile = tile + gx; // narrow: static round-robin
        // wide: tiles are drawn from a counter (rows differ a lot in length) -- except in slot-major order, where the workgroups of a slot start
        // far apart in time and the early ones would draw most of the tiles into THEIR sub-buckets (measured: bucket overflow, call re-run)
        if (WIDE && !slot_major && threadIdx.x == 0) // asked for now, needed after the tile: the atomic's latency is hidden
            next_tile = gx + atomicAdd(&d.tile_ctr[d.launch_par][q * CSTRIDE], 1u);
        // ---- one frontier entry per lane
        const uint32_t i = tile_pos<GRAN>(threadIdx.x, tile, seg_len);
        uint32_t cnt = 0;
        if (i < count) {
            const uint32_t v = in[i];
            uint64_t inc = incs[i]; // first pass: the residue taken from v; later passes: its increment
            const uint64_t ri = QUAD ? d.rowinfo4[v] : d.rowinfo[v];
            int64_t beg = (int64_t)(ri >> 24); // QUAD: the row's first quad
            uint32_t deg = QUAD ? (((uint32_t)ri & DEG_SAT) == DEG_SAT ? (uint32_t)(d.row_ptr[v + 1] - d.row_ptr[v]) : (uint32_t)ri & DEG_SAT) : ri_deg(d, ri, v);
            if (first_pass) { // pop (algo.h:983-1002): the residue word itself was zeroed when v entered the list
                const uint64_t rsv_old = d.ppr[slab + v];
                uint64_t rsv_add, dang;
                inc = pop_value(d.afix, inc, deg, rsv_add, dang);
                incs[i] = inc;
                if (rsv_add) d.ppr[slab + v] = rsv_old + rsv_add; // algo.h:986-989 (only this lane owns (q, v))
                acc_res += rsv_add; acc_dang += dang; acc_pops++; acc_relax += deg;
            }
            if (SPLIT && d.row_split) { // only the part of the (sorted) row whose targets belong to this pass
                const uint32_t *sp = d.row_split + (uint64_t)v * (d.npass + 1) + d.pass;
                beg += sp[0];
                deg = sp[1] - sp[0];
            }
            s_ebeg[threadIdx.x] = beg;
            s_inc[threadIdx.x] = inc;
            cnt = inc ? deg : 0u; // an increment of zero changes nothing: skip the row
            if (QUAD) cnt = (cnt + 3u) >> 2; // the tile's rows are concatenated quad by quad
        }
        uint32_t total;
        const uint32_t pre = block_excl_scan_n<NT>(cnt, s_w, total);
        s_pref[threadIdx.x] = pre;
        if (threadIdx.x == 0) s_pref[NT] = total;
        __syncthreads();
        STAMP(0);
        // ---- bin the tile's edges
        constexpr uint32_t UNITS = QUAD ? CHUNK / 4 : CHUNK; // edges (quads) of the concatenated rows a chunk takes
        for (uint32_t cb = 0; cb < total; cb += UNITS) {
            uint32_t w[BIN_EPT], rank[BIN_EPT], si[BIN_EPT];
            if (QUAD) {
                // lane t takes BIN_EPT / 4 consecutive QUADS: one binary search, one step and ONE 16-byte load per quad.
                // (Round 6, measured and dropped: the loads of chunk c + 1 issued as soon as chunk c's quads are unpacked, so that they
                // fly while chunk c is ranked, staged and written out -- LJ-sized 304.7 / 313.0 against 310.9 / 311.1 ms, Twitter-2010-sized
                // 594 / 587 against 586 / 587, 17 VGPR spills in <2560>: the other waves already cover that latency; what one more load per
                // quad costs (+ 25 % / + 17 %, profiles/r06_wide_probes.txt) is memory-system throughput for short random rows.)
                constexpr int QEPT = BIN_EPT / 4;
                static_assert(!QUAD || BIN_EPT % 4 == 0, "whole quads per lane");
                const uint32_t q0 = cb + threadIdx.x * QEPT;
                uint32_t lo = 0;
                if (q0 < total) {
                    uint32_t hi = NT;
#pragma unroll
                    for (int it = 0; it < (NT == 256 ? 8 : NT == 512 ? 9 : 10); it++) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_pref[mid] <= q0) lo = mid; else hi = mid;
                    }
                }
                uint32_t sq[QEPT > 0 ? QEPT : 1];
#pragma unroll
                for (int j = 0; j < QEPT; j++) { // entries without edges: step over them
                    const uint32_t qq = q0 + j;
                    if (qq < total) while (s_pref[lo + 1] <= qq) lo++;
                    sq[j] = lo;
                }
                uint4 x[QEPT > 0 ? QEPT : 1], xp[QEPT > 0 ? QEPT : 1];
#pragma unroll
                for (int j = 0; j < QEPT; j++) { // straight-line: all loads in flight together
                    const uint32_t qq = q0 + j;
                    x[j] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                    xp[j] = make_uint4(0u, 0u, 0u, 0u);
                    if (qq < total) {
                        const int64_t at = s_ebeg[sq[j]] + (qq - s_pref[sq[j]]);
                        x[j] = ((const uint4 *)colsrc)[at];
                        xp[j] = diag::bin_load_probe(colsrc, d.col4, d.col_hub4, d.col, at);
                    }
                }
#pragma unroll
                for (int j = 0; j < QEPT; j++) {
                    diag::bin_load_keep(xp[j]);
                    w[4 * j] = x[j].x; w[4 * j + 1] = x[j].y; w[4 * j + 2] = x[j].z; w[4 * j + 3] = x[j].w; // (padding words are 0xFFFFFFFF: no message)
                    si[4 * j] = si[4 * j + 1] = si[4 * j + 2] = si[4 * j + 3] = sq[j];
                }
            } else {
            const uint32_t e0 = cb + threadIdx.x * BIN_EPT; // lane t takes 8 consecutive edges: ONE binary search for the source entry
            uint32_t lo = 0;
            if (e0 < total) {
                uint32_t hi = NT;
#pragma unroll
                for (int it = 0; it < (NT == 256 ? 8 : NT == 512 ? 9 : 10); it++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_pref[mid] <= e0) lo = mid; else hi = mid;
                }
            }
#pragma unroll
            for (int k = 0; k < BIN_EPT; k++) { // entries without edges: step over them
                const uint32_t e = e0 + k;
                if (e < total) while (s_pref[lo + 1] <= e) lo++;
                si[k] = lo;
            }
#pragma unroll
            for (int k = 0; k < BIN_EPT; k++) { // straight-line: all BIN_EPT gathers in flight together
                const uint32_t e = e0 + k;
                w[k] = 0xFFFFFFFFu;
                if (e < total) w[k] = (uint32_t)colsrc[s_ebeg[si[k]] + (e - s_pref[si[k]])];
            }
            }
            if (hubmode) {
#pragma unroll
                for (int k = 0; k < BIN_EPT; k++)
                    if (w[k] != 0xFFFFFFFFu && (w[k] & 0x80000000u)) { // a hub: summed here, one row of sums per workgroup and level
                        atomicAdd(&s_hub[w[k] & 0x7FFFFFFFu], (unsigned long long)s_inc[si[k]]);
                        w[k] = 0xFFFFFFFFu;
                    }
            }
#pragma unroll
            for (int k = 0; k < BIN_EPT; k++) {
                if (w[k] != 0xFFFFFFFFu && (w[k] >> BS) - bin_lo >= bin_cnt) w[k] = 0xFFFFFFFFu; // another pass's bins
                if (w[k] != 0xFFFFFFFFu) rank[k] = atomicAdd(&s_cnt[(w[k] >> BS) - bin_lo], 1u); // rank inside (chunk, bin)
                if (w[k] != 0xFFFFFFFFu) BIN_RANK_PROBE((w[k] >> BS) - bin_lo);
            }
            diag::bin_sync_probe();
            __syncthreads();
            STAMP(1);
            uint32_t staged; // messages of this chunk that belong to the pass's bins
            { // take sub-bucket space (a counter in LDS, no atomic) and lay the bins out in the LDS stage:
              // lane t owns bins t*PER .. t*PER+PER-1
                constexpr int PER = (NB + NT - 1) / NT;
                uint32_t c[PER], mine = 0;
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const uint32_t b = threadIdx.x * PER + j;
                    c[j] = b < (uint32_t)d.bin_cnt && b < (uint32_t)NB ? s_cnt[b] : 0;
                    mine += c[j];
                }
                uint32_t ctot;
                uint32_t pre2 = block_excl_scan_n<NT>(mine, s_w, ctot);
                staged = ctot;
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const uint32_t b = threadIdx.x * PER + j;
                    if (b < (uint32_t)NB) {
                        s_lofs[b] = pre2;
                        pre2 += c[j];
                        if (c[j]) {
                            s_fill[b] += (c[j] + (PAD - 1)) & ~(uint32_t)(PAD - 1);
                            s_cnt[b] = 0;
                        }
                    }
                }
                if (threadIdx.x == NT - 1) s_lofs[NB] = ctot;
            }
            __syncthreads();
            STAMP(2);
#pragma unroll
            for (int k = 0; k < BIN_EPT; k++) {
                if (w[k] != 0xFFFFFFFFu) {
                    const uint32_t b = (w[k] >> BS) - bin_lo;
                    const uint32_t sp = s_lofs[b] + rank[k];
                    const uint32_t own = si[k];
                    const uint32_t sw = WIDE ? (w[k] & (BSZ - 1)) | (own << BS) : ((w[k] & (BSZ - 1)) << SEG_BITS) | tile_pos<GRAN>(own, tile, seg_len);
                    s_msg[sp] = sw;
                    diag::bin_lds_write_probe(s_msg, sp, sw);
                    s_bin[sp] = b;
                }
            }
            __syncthreads();
            STAMP(3);
            for (uint32_t m = threadIdx.x; m < staged; m += NT) { // consecutive lanes -> consecutive bucket slots
                const uint32_t e = s_msg[m];
                diag::bin_lds_read_probe(s_msg, m);
                const uint32_t b = s_bin[m];
                uint32_t sidx = 0, local;
                if (WIDE) { sidx = e >> BS; local = e & (BSZ - 1); }
                else local = e >> SEG_BITS; // narrow: the word names the frontier position
                const uint32_t crun = s_lofs[b + 1] - s_lofs[b]; // messages of this (chunk, bin) run
                const uint32_t pos = s_fill[b] - ((crun + (PAD - 1)) & ~(uint32_t)(PAD - 1)) + (m - s_lofs[b]); // s_fill already counts this chunk's (padded) run
                const uint64_t at = bk0 + (uint64_t)b * bstride + pos;
                bool parked = pos >= d.bk_cap; // sub-bucket full
                if (!parked) {
                    if (WIDE) {
                        const uint64_t inc = s_inc[sidx];
                        parked = inc >= WIDE_MAXV; // does not fit the packed word: null word here, the increment goes to the list
                        const uint64_t mword = parked ? 0ull : (uint64_t)local | (inc << BS);
                        d.bk_inc[at] = mword;
                        diag::bin_store_probe(&d.bk_inc[at], mword);
                    } else d.bk_w[at] = e;
                }
                if (parked) { // park the increment in the slot's overflow list, folded in by k_accum
                    if (!WIDE) { // back from the frontier position to the entry inside the tile (inverse of tile_pos)
                        const uint32_t fp = e & ((1u << SEG_BITS) - 1), sg = fp / seg_len;
                        sidx = sg * GRAN + (fp - sg * seg_len - tile * GRAN);
                    }
                    const uint32_t oi = atomicAdd(&d.ov_count[par][q * CSTRIDE], 1u);
                    if (oi < d.ov_cap) {
                        d.ov_w[(uint64_t)q * d.ov_cap + oi] = ((bin_lo + b) << BS) | local;
                        d.ov_inc[(uint64_t)q * d.ov_cap + oi] = s_inc[sidx];
                        atomicAdd(&d.ov_bin[par][(uint64_t)q * d.nbins + bin_lo + b], 1u);
                    } else atomicOr(d.err, ERR_BUCKET_OVERFLOW);
                }
            }
            if (PAD > 1) { // null words up to the sector boundary (merged with the run's last sector in L2)
                for (uint32_t b = threadIdx.x; b < bin_cnt; b += NT) {
                    const uint32_t crun = s_lofs[b + 1] - s_lofs[b];
                    const uint32_t prun = (crun + (PAD - 1)) & ~(uint32_t)(PAD - 1);
                    for (uint32_t i = crun; i < prun; i++) {
                        const uint32_t pos = s_fill[b] - prun + i;
                        if (pos < d.bk_cap) d.bk_inc[bk0 + (uint64_t)b * bstride + pos] = 0ull;
                    }
                }
            }
            STAMP(4);
        }
        if (WIDE && threadIdx.x == 0) s_next = next_tile;
        __syncthreads();
        tile = WIDE ? s_next : next_tile; // (slot-major: thread 0's next_tile is the static one)
        STAMP(5);
    }
    for (uint32_t i = threadIdx.x; i < bin_cnt; i += NT) bkc[(uint64_t)i * sub] = s_fill[i];
    if (hubmode) { // the workgroup's row of hub sums (every workgroup of the slot writes one, with or without tiles: k_accum reads them all)
        __syncthreads();
        uint64_t *row = d.hubsum + ((uint64_t)q * sub + bx) * d.hubs;
        for (uint32_t i = threadIdx.x; i < d.hubs; i += NT) row[i] = s_hub[i];
    }
    STAMP_FLUSH(0);
    if (!first_pass) return; // only the first pass pops
    // the slot's counters share one 128-byte line and one address takes ~20 M atomics/s: add them up over the workgroup
    // first (Twitter-2010-sized: 128 workgroups x 16 waves x 4 counters per slot and level otherwise)
    acc_res = wave_sum(acc_res); acc_dang = wave_sum(acc_dang);
    acc_pops = wave_sum(acc_pops); acc_relax = wave_sum(acc_relax);
    __syncthreads(); // s_inc is free after the last tile
    if (lane == 0) {
        const int w = threadIdx.x >> 6;
        s_inc[w * 4 + 0] = acc_res; s_inc[w * 4 + 1] = acc_dang; s_inc[w * 4 + 2] = acc_pops; s_inc[w * 4 + 3] = acc_relax;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) t += s_inc[w * 4 + threadIdx.x];
        QState *qs = &d.qs[q];
        unsigned long long *dst = threadIdx.x == 0 ? &qs->reserved : threadIdx.x == 1 ? (unsigned long long *)&qs->dang[par]
                                  : threadIdx.x == 2 ? &qs->pops : &qs->relax; // rsum bookkeeping, algo.h:992
        if (t) atomicAdd(dst, (unsigned long long)t);
    }
}

// ---- tail of the push.  Once every slot's frontier is small the per-level launches are pure latency (two launches of
// ~10-20 us each plus their gaps, for a few hundred relaxations per slot).  This kernel finishes the push of a slot
// inside ONE workgroup: the same level-synchronous schedule -- all pops of a level, then its relaxations, then the
// dangling mass -- with workgroup barriers instead of launches and one returning atomic per relaxation (the slot's
// slabs are touched by this workgroup only; every mutable word is read and written through atomics or cache-bypassing
// loads, so no stale L1 line is ever observed).  Integer adds commute: same bits as the bucketed levels and the twin.
// grid = nq, TAIL_THREADS threads.  L0: first level to run (its entries carry their residue, see k_accum);
// max_levels: stop after that many more (0: until empty).
#ifndef FORA_TAIL_THREADS
#define FORA_TAIL_THREADS 1024
#endif
#ifndef FORA_TAIL_EPT
#define FORA_TAIL_EPT 4
#endif
#ifndef FORA_TAIL_WPE
#define FORA_TAIL_WPE 4
#endif
constexpr int TAIL_THREADS = FORA_TAIL_THREADS;
constexpr int TAIL_EPT = FORA_TAIL_EPT; // relaxations a lane keeps in flight
__global__ void __launch_bounds__(TAIL_THREADS, FORA_TAIL_WPE) k_push_tail(Dev d, int L0, int max_levels) {
    __shared__ int64_t s_ebeg[TAIL_THREADS];
    __shared__ uint64_t s_inc[TAIL_THREADS];
    __shared__ uint32_t s_pref[TAIL_THREADS + 1];
    __shared__ uint32_t s_scan[TAIL_THREADS / 64];
    __shared__ uint32_t s_next, s_count, s_ndue, s_nwait, s_real;
    __shared__ unsigned long long s_dang;
    // Round 6: a small level is a chain of dependent round trips (profiles/r06_tail.txt: nine of them per level, one workgroup per
    // slot, nothing to overlap them with).  A level of at most TAIL_THREADS nodes now (a) takes its list from LDS (s_front: the
    // crossing nodes of the level before, also stored to the slot's global list as before), (b) pops in the tile prologue -- residue
    // exchange, row word and degree of a node in ONE round trip, the increment stays in LDS instead of going through inc_tab --,
    // (c) knows its size from LDS, and (d) the hubs' node ids / degrees sit in registers for the launch: four round trips per level.
    __shared__ uint32_t s_front[2][TAIL_THREADS];
    constexpr uint32_t HOLE = 0x80000000u; // list entry of a node that crossed but waits a level (bounded deferral): skipped by the pops
    // The chip takes 23 G returning atomics per second whatever their addresses (profiles/r01_atomics_microbench.txt), and
    // this kernel issues little else: with a hub copy of the graph (Dev::col_hub) the increments for the hubs -- 45 % of all
    // relaxations on the ws-sized graph -- are summed in LDS and reach the residue as one atomic per touched hub and level.
    extern __shared__ unsigned long long s_hubt[]; // [d.hubs] when hubmode
    const bool hubmode = d.tail_hubs && d.col_hub;
    const int32_t *colsrc = hubmode ? d.col_hub : d.col;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t src = (uint32_t)d.src[q];
    const int dk = (!TEST_PATHS || max_levels > 0) ? 0 : d.defer_k; // capped runs (power iteration) keep plain levels
    if (hubmode) for (uint32_t h = tid; h < d.hubs; h += TAIL_THREADS) s_hubt[h] = 0;
    constexpr int HPT = 4; // hubs per thread whose node id and threshold-degree ride in registers (hubs beyond HPT * TAIL_THREADS: loaded per level)
    uint32_t hub_w[HPT], hub_dg[HPT];
#pragma unroll
    for (int k = 0; k < HPT; k++) {
        const uint32_t h = k * TAIL_THREADS + tid;
        hub_w[k] = hubmode && h < d.hubs ? d.hub_node[h] : 0u;
        hub_dg[k] = hubmode && h < d.hubs ? d.deg[hub_w[k]] : 0u;
    }
    bool list_in_lds = false; // the level's list is complete in s_front[L & 1] (wave-uniform, same in every thread)
    uint64_t acc_res = 0, acc_pops = 0, acc_relax = 0;
    int L = L0;
    uint32_t levels_run = 0;
    if (tid == 0) { s_ndue = 0; s_real = __atomic_load_n(&d.fl_count[L0 & 1][q * CSTRIDE], __ATOMIC_RELAXED); }
    __syncthreads();
    if (dk && __atomic_load_n(&d.fl_count[L0 & 1][q * CSTRIDE + 2], __ATOMIC_RELAXED)) {
        // nodes the last bucketed level deferred (marks in the slot's bitmap, see Dev::dbm): they are due in this kernel's
        // first level; from here on the deferred sets are lists
        uint64_t *bm = d.dbm[L0 & 1] + (uint64_t)q * d.dbm_words;
        uint32_t *due0 = d.dl[L0 & 1] + slab;
        for (uint32_t w = tid; w < d.dbm_words; w += TAIL_THREADS) {
            uint64_t wd = bm[w];
            if (!wd) continue;
            bm[w] = 0;
            while (wd) {
                const uint32_t bit = (uint32_t)__ffsll((long long)wd) - 1;
                due0[atomicAdd(&s_ndue, 1u)] = (w << 6) + bit;
                wd &= wd - 1;
            }
        }
        for (uint32_t b = tid; b < (uint32_t)d.nbins; b += TAIL_THREADS) d.dflag[L0 & 1][(uint64_t)q * d.nbins + b] = 0;
        __syncthreads();
    }
    // Visibility inside the workgroup: every mutable word is read and written with atomics or L1-bypassing loads, plain
    // stores (frontier lists, increments) are complete at the next __syncthreads() -- no agent-scope fence is needed
    // (three __threadfence() per level cost more than the level's work on small frontiers).
    for (int done = 0; max_levels <= 0 || done < max_levels; done++, L++) {
        const int par = L & 1;
        uint64_t *incs = d.inc_tab[par] + (uint64_t)q * d.segq_cap;
        if (tid == 0) {
            if (done == 0) s_count = __atomic_load_n(&d.fl_count[par][q * CSTRIDE], __ATOMIC_RELAXED); // (later levels: set at the end of the level before)
            s_next = 0;
            s_nwait = 0;
            s_dang = 0;
        }
        __syncthreads();
        const uint32_t count = s_count, real = s_real, ndue = s_ndue;
        if (!real && !ndue) break;
        if (L >= MAX_LEVELS) { if (tid == 0) atomicOr(d.err, ERR_WL_OVERFLOW); break; }
        const uint32_t *in = d.fl[par] + slab;
        uint32_t *out = d.fl[par ^ 1] + slab;
        // ---- all pops of the level (algo.h:983-1002).  The entries of level L0 carry the residue already taken from
        // their nodes (see k_accum); later levels are collected by this kernel and take it here.
        if (real) levels_run++;
        const bool fused = count <= (uint32_t)TAIL_THREADS && !dk; // one tile: the pops happen in its prologue (no holes without deferral)
        uint32_t *nxt = s_front[par ^ 1];
        for (uint32_t i = tid; !fused && i < count; i += TAIL_THREADS) {
            const uint32_t v = __atomic_load_n(&in[i], __ATOMIC_RELAXED);
            if (v & HOLE) continue;
            const uint64_t a = slab + v;
            const uint64_t r = done ? atomicExch((unsigned long long *)&d.residue[a], 0ull)      // algo.h:984-985
                                    : __atomic_load_n(&incs[i], __ATOMIC_RELAXED);
            const uint32_t deg = d.deg[v];
            uint64_t res_add, dang;
            const uint64_t inc = pop_value(d.afix, r, deg, res_add, dang);                  // algo.h:986-1002
            if (dang) atomicAdd(&s_dang, (unsigned long long)dang);                         // algo.h:993-994
            atomicAdd((unsigned long long *)&d.ppr[a], (unsigned long long)res_add);        // algo.h:986-989
            incs[i] = inc;
            acc_res += res_add;
            acc_pops++;
            acc_relax += deg;
        }
        __syncthreads();
        // ---- relaxations, a tile of TAIL_THREADS frontier nodes at a time (algo.h:1003-1016); a lane takes TAIL_EPT
        // consecutive edges of the tile and keeps their gathers, then their adds, in flight together
        for (uint32_t tbase = 0; tbase < count; tbase += TAIL_THREADS) {
            const uint32_t i = tbase + tid;
            uint32_t cnt = 0;
            if (fused) {
                if (i < count) { // the pop (algo.h:983-1002) and the row of the node in one round trip
                    const uint32_t v = list_in_lds ? s_front[par][i] : __atomic_load_n(&in[i], __ATOMIC_RELAXED);
                    const uint64_t a = slab + v;
                    const uint64_t r = done ? atomicExch((unsigned long long *)&d.residue[a], 0ull)      // algo.h:984-985
                                            : __atomic_load_n(&incs[i], __ATOMIC_RELAXED);
                    int64_t beg; uint64_t deg;
                    node_row(d, v, beg, deg);
                    uint64_t res_add, dang;
                    const uint64_t inc = pop_value(d.afix, r, (uint32_t)deg, res_add, dang);             // algo.h:986-1002
                    if (dang) atomicAdd(&s_dang, (unsigned long long)dang);                               // algo.h:993-994
                    atomicAdd((unsigned long long *)&d.ppr[a], (unsigned long long)res_add);              // algo.h:986-989
                    acc_res += res_add;
                    acc_pops++;
                    acc_relax += deg;
                    s_ebeg[tid] = beg;
                    s_inc[tid] = inc;
                    cnt = inc ? (uint32_t)deg : 0u;
                }
            } else if (i < count) {
                const uint32_t v = __atomic_load_n(&in[i], __ATOMIC_RELAXED);
                if (!(v & HOLE)) {
                    int64_t beg; uint64_t deg;
                    node_row(d, v, beg, deg);
                    s_ebeg[tid] = beg;
                    const uint64_t inc = __atomic_load_n(&incs[i], __ATOMIC_RELAXED);
                    s_inc[tid] = inc;
                    cnt = inc ? (uint32_t)deg : 0u;
                }
            }
            uint32_t wtot;
            const uint32_t wx = wave_excl_scan(cnt, wtot);
            if (lane == 0) s_scan[wid] = wtot;
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < TAIL_THREADS / 64; w++) { const uint32_t c = s_scan[w]; if (w < wid) before += c; total += c; }
            s_pref[tid] = before + wx;
            if (tid == 0) s_pref[TAIL_THREADS] = total;
            __syncthreads();
            for (uint32_t cb = 0; cb < total; cb += TAIL_THREADS * TAIL_EPT) {
                const uint32_t e0 = cb + tid * TAIL_EPT;
                uint32_t lo = 0;
                if (e0 < total) {
                    uint32_t hi = TAIL_THREADS;
#pragma unroll
                    for (int it = 0; it < (TAIL_THREADS == 1024 ? 10 : TAIL_THREADS == 512 ? 9 : 8); it++) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_pref[mid] <= e0) lo = mid; else hi = mid;
                    }
                }
                uint32_t w[TAIL_EPT], dg[TAIL_EPT];
                uint64_t inc[TAIL_EPT], old[TAIL_EPT];
#pragma unroll
                for (int k = 0; k < TAIL_EPT; k++) {
                    const uint32_t e = e0 + k;
                    w[k] = 0xFFFFFFFFu; inc[k] = 0;
                    if (e < total) {
                        while (s_pref[lo + 1] <= e) lo++; // nodes without edges
                        inc[k] = s_inc[lo];
                        w[k] = (uint32_t)colsrc[s_ebeg[lo] + (e - s_pref[lo])];
                    }
                }
                if (hubmode) {
#pragma unroll
                    for (int k = 0; k < TAIL_EPT; k++)
                        if (w[k] != 0xFFFFFFFFu && (w[k] & 0x80000000u)) { // a hub: summed here, added to its residue at the end of the level
                            atomicAdd(&s_hubt[w[k] & 0x7FFFFFFFu], (unsigned long long)inc[k]);
                            w[k] = 0xFFFFFFFFu;
                        }
                }
#pragma unroll
                for (int k = 0; k < TAIL_EPT; k++) {
                    old[k] = 0; dg[k] = 0;
                    if (w[k] != 0xFFFFFFFFu) {
                        old[k] = atomicAdd((unsigned long long *)&d.residue[slab + w[k]], (unsigned long long)inc[k]);
                        dg[k] = d.deg[w[k]];
                    }
                }
#pragma unroll
                for (int k = 0; k < TAIL_EPT; k++) {
                    if (w[k] != 0xFFFFFFFFu) {
                        const uint64_t thr = node_thr(d.t1, dg[k]);
                        if (old[k] < thr && old[k] + inc[k] >= thr) { // crossed in this level: next frontier (algo.h:1012-1015)
                            const uint32_t pos = atomicAdd(&s_next, 1u);
                            if (pos < (uint32_t)d.n) out[pos] = w[k]; else atomicOr(d.err, ERR_WL_OVERFLOW);
                            if (pos < (uint32_t)TAIL_THREADS) nxt[pos] = w[k];
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (hubmode) { // the hubs' sums of the level (every add to s_hubt is behind the tile loop's last barrier)
            uint32_t k = 0;
            for (uint32_t h = tid; h < d.hubs; h += TAIL_THREADS, k++) {
                const uint64_t hv = s_hubt[h];
                if (!hv) continue;
                s_hubt[h] = 0;
                uint32_t w, wdg;
                if (k < (uint32_t)HPT) { // (registers; the compiler unrolls the select over the HPT pairs)
                    w = hub_w[0]; wdg = hub_dg[0];
#pragma unroll
                    for (int kk = 1; kk < HPT; kk++) if (k == (uint32_t)kk) { w = hub_w[kk]; wdg = hub_dg[kk]; }
                } else { w = d.hub_node[h]; wdg = d.deg[w]; }
                const uint64_t old = atomicAdd((unsigned long long *)&d.residue[slab + w], (unsigned long long)hv);
                const uint64_t thr = node_thr(d.t1, wdg);
                if (old < thr && old + hv >= thr) { // (algo.h:1012-1015: crossed in this level)
                    const uint32_t pos = atomicAdd(&s_next, 1u);
                    if (pos < (uint32_t)d.n) out[pos] = w; else atomicOr(d.err, ERR_WL_OVERFLOW);
                    if (pos < (uint32_t)TAIL_THREADS) nxt[pos] = w;
                }
            }
        }
        // ---- dangling mass back to the source (algo.h:994-998)
        if (tid == 0 && s_dang) {
            const uint64_t dm = s_dang;
            const uint64_t old = atomicAdd((unsigned long long *)&d.residue[slab + src], (unsigned long long)dm);
            const uint64_t thr = node_thr(d.t1, d.deg[src]);
            if (old < thr && old + dm >= thr) {
                const uint32_t pos = atomicAdd(&s_next, 1u);
                if (pos < (uint32_t)d.n) out[pos] = src; else atomicOr(d.err, ERR_WL_OVERFLOW);
                if (pos < (uint32_t)TAIL_THREADS) nxt[pos] = src;
            }
        }
        __syncthreads();
        uint32_t crossed = min(s_next, (uint32_t)d.n);
        list_in_lds = !dk && crossed <= (uint32_t)TAIL_THREADS; // (every entry of the next list went to s_front too)
        if (dk) {
            const int dkl = real >= d.defer_min ? dk : 0; // this level defers only if it popped enough nodes
            // bounded deferral (see Dev::dbm): every add of the level has landed; a node that only just crossed becomes a
            // hole in the next list and waits in dl[par ^ 1]; the nodes that waited since the previous level join the list
            uint32_t *wait = d.dl[par ^ 1] + slab;
            const uint32_t *due = d.dl[par] + slab;
            for (uint32_t i = tid; i < crossed; i += TAIL_THREADS) {
                const uint32_t w = out[i];
                const uint64_t r = __atomic_load_n(&d.residue[slab + w], __ATOMIC_RELAXED);
                if (dkl && (r >> dkl) < node_thr(d.t1, d.deg[w])) {
                    out[i] = w | HOLE;
                    wait[atomicAdd(&s_nwait, 1u)] = w;
                }
            }
            for (uint32_t i = tid; i < ndue; i += TAIL_THREADS) {
                const uint32_t pos = crossed + i;
                if (pos < (uint32_t)d.n) out[pos] = __atomic_load_n(&due[i], __ATOMIC_RELAXED); else atomicOr(d.err, ERR_WL_OVERFLOW);
            }
            __syncthreads();
        }
        if (tid == 0) {
            const uint32_t nwait = dk ? s_nwait : 0u;
            s_count = min(crossed + (dk ? ndue : 0u), (uint32_t)d.n); // the next level's list size (also in the slot's global word: a capped run ends here)
            d.fl_count[par ^ 1][q * CSTRIDE] = s_count;
            d.fl_count[par][q * CSTRIDE] = 0;
            s_real = crossed - nwait + (dk ? ndue : 0u);
            s_ndue = nwait;
        }
        __syncthreads();
    }
    acc_res = wave_sum(acc_res); acc_pops = wave_sum(acc_pops); acc_relax = wave_sum(acc_relax);
    if (lane == 0 && acc_pops) {
        QState *s = &d.qs[q];
        atomicAdd(&s->reserved, (unsigned long long)acc_res);
        atomicAdd(&s->pops, (unsigned long long)acc_pops);
        if (acc_relax) atomicAdd(&s->relax, (unsigned long long)acc_relax);
    }
    if (tid == 0 && levels_run) d.qs[q].levels += levels_run;
}

// grid = (bins of the pass, nq), ACC_THREADS (wide layouts: ACC_THREADS_WIDE) threads.  TO_PPR: the buckets hold walk results; they are added
// to the ppr slab and there is no threshold / frontier.
//
// Push form (TO_PPR = false), level L: after the sweep has added the level's increments to the residue range this
// workgroup owns, every node that crossed its threshold (algo.h:1012) enters the frontier of level L + 1 right here
// (d.pop_next = 0 on the last level of a capped run leaves it alone): its residue is in registers, so the slab word is
// stored as 0 (algo.h:984-985) and (node, residue) is appended at a rank that follows the NODE ORDER inside the bin.
// The bin kernel of the next level finishes the pop with dense lanes and never gathers a residue.
// (The argument struct is read through the kernarg segment -- constant address space: scalar loads -- so that the wide kernel can launder the
// pointer per bin: by value, the loop over a workgroup's bins hoisted the ~40 fields the body reads into SGPRs for the whole launch, 39 of them spilled.)
typedef const __attribute__((address_space(4))) Dev &DevRef;
__device__ __forceinline__ const __attribute__((address_space(4))) Dev *dev_args() {
    auto p = (const __attribute__((address_space(4))) Dev *)__builtin_amdgcn_kernarg_segment_ptr(); // (Dev is the kernel's first argument)
    asm volatile("" : "+s"(p));
    return p;
}
template <bool TO_PPR, bool WIDE>
__device__ __forceinline__ void accum_bin(DevRef d, int L, const int lb, const int q) { // lb: bin inside the pass
    constexpr int BS = WIDE ? BIN_SHIFT_WIDE : BIN_SHIFT; // bits of a local target
    constexpr uint32_t BSZ = 1u << BS;
    constexpr int AT = WIDE ? ACC_THREADS_WIDE : ACC_THREADS;
    __shared__ uint64_t acc[BSZ];
    constexpr int NW = AT / 64;
    constexpr int SWEEP = BSZ / AT;
    static_assert(SWEEP <= 32, "crossmask holds one bit per swept node of a lane");
    __shared__ uint32_t s_rank[SWEEP * NW + 1];
    __shared__ uint32_t s_gbase;
    __shared__ uint32_t s_list[TO_PPR ? 1 : 1024];
    __shared__ uint32_t s_nlist;
    __shared__ uint32_t s_scnt[MAX_SUB], s_total, s_ovn, s_din, s_ndef;
    const int b = d.bin_lo + lb; // bin of the graph
    const int par = L & 1;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t sub = d.sub;
    const uint64_t bi = (uint64_t)q * d.pbins + lb;
    // the bucket's sub-buckets: counts (clamped: the excess is in the overflow list (push) / went by direct atomics (walks))
    if (threadIdx.x < 64) {
        uint32_t t = 0;
        for (uint32_t x = threadIdx.x; x < sub; x += 64) {
            uint32_t c = d.bk_count[bi * sub + x];
            if (c) d.bk_count[bi * sub + x] = 0;
            if (c > d.bk_cap) c = d.bk_cap;
            s_scnt[x] = c;
            t += c;
        }
        t = (uint32_t)wave_sum((uint64_t)t);
        if (threadIdx.x == 0) {
            s_total = t; s_nlist = 0;
            uint32_t o = 0;
            if (!TO_PPR && d.ov_bin[par][(uint64_t)q * d.nbins + b]) { // a hub's bin at a dense level; usually none
                d.ov_bin[par][(uint64_t)q * d.nbins + b] = 0;
                o = d.ov_count[par][q * CSTRIDE];
                if (o > d.ov_cap) o = d.ov_cap;
            }
            s_ovn = o;
            uint32_t di = 0;
            if (TEST_PATHS && !TO_PPR && d.defer_k) { // nodes of this bin deferred by the previous level: they are due now
                di = d.dflag[par][(uint64_t)q * d.nbins + b];
                if (di) d.dflag[par][(uint64_t)q * d.nbins + b] = 0;
            }
            s_din = di; s_ndef = 0;
        }
    }
    const uint32_t s = (uint32_t)d.src[q];
    const uint64_t dm = (!TO_PPR && (int)(s >> BS) == b) ? (uint64_t)d.qs[q].dang[par] : 0; // algo.h:994
    uint64_t *target = TO_PPR ? d.ppr : d.residue;
    const uint32_t *xl = TO_PPR ? d.acc_xl : nullptr; // walk results in WalkDG bucket order: original id = xl[bin << BS | local]
    __syncthreads();
    const uint32_t cnt = s_total;
    if (threadIdx.x == 0 && dm) d.qs[q].dang[par] = 0;
    const uint32_t ovn = s_ovn; // entries of the slot's overflow list to scan: 0 unless some belong to THIS bin
    const bool din = TEST_PATHS && !TO_PPR && s_din != 0;
    // hub pre-aggregation (Dev::col_hub): the level's bin kernel summed the increments of this bin's hubs per workgroup
    const bool hubmode = !TO_PPR && d.col_hub && d.fl_count[par][q * CSTRIDE] >= d.hub_min;
    const uint32_t hub_lo = hubmode ? d.hub_first[b] : 0, hub_hi = hubmode ? d.hub_first[b + 1] : 0;
    if (cnt == 0 && dm == 0 && ovn == 0 && !din && hub_hi == hub_lo) return;
    const int dk = (!TEST_PATHS || TO_PPR || !d.defer_k) ? 0 : (d.fl_count[par][q * CSTRIDE] >= d.defer_min ? d.defer_k : 0); // (the level's own frontier size: nothing writes it during the level)
    const uint32_t wpb = BSZ / 64; // bitmap words per bin
    uint64_t *dbm_in = (!TEST_PATHS || TO_PPR) ? nullptr : d.dbm[par] + (uint64_t)q * d.dbm_words + (uint64_t)b * wpb;
    uint64_t *dbm_out = (!TEST_PATHS || TO_PPR) ? nullptr : d.dbm[par ^ 1] + (uint64_t)q * d.dbm_words + (uint64_t)b * wpb;
    uint32_t *fl_next = d.fl[par ^ 1] + slab;
    uint64_t *inc_next = TO_PPR ? nullptr : d.inc_tab[par ^ 1] + (uint64_t)q * d.segq_cap;
    uint32_t *flc_next = &d.fl_count[par ^ 1][q * CSTRIDE];
    const uint64_t bk0 = bi * sub * d.bk_cap; // sub-bucket x starts at bk0 + x * bk_cap
    const uint32_t node0 = (uint32_t)b << BS;
    const uint64_t *itab = TO_PPR ? nullptr : d.inc_tab[par] + (uint64_t)q * d.segq_cap;
    const uint64_t t1q = TO_PPR ? 0 : thr_unit(d.t1, d.qs[q].tshift); // the slot's threshold unit in its current round
    // message forms: narrow push = 4-byte word (local target << SEG_BITS | frontier position), increment gathered from the
    // table; everything else = ONE 64-bit word in bk_inc: narrow walk results node id | weight << WPACK_SHIFT, wide
    // messages local target | value << BS
    const bool gather = !TO_PPR && !WIDE;
    const bool packed = !gather;
    const int pshift = WIDE ? BS : WPACK_SHIFT;
    if (ovn == 0 && !din && hub_hi == hub_lo && cnt + (dm ? 1 : 0) <= d.tiny_max) {
        // small bucket: zeroing and sweeping 64 KiB of LDS would cost more than its atomics (the workgroup owns the
        // node range and the level's pops are done, so nothing else touches these words).  Wave w takes sub-buckets
        // w, w + NW, ...
        for (uint32_t x = wid; x <= sub; x += NW) {
            const uint32_t n_x = x < sub ? s_scnt[x] : (dm ? 1u : 0u); // x == sub: the dangling mass, one more "message"
            const uint64_t at0 = bk0 + (uint64_t)x * d.bk_cap;
            for (uint32_t i0 = 0; i0 < n_x; i0 += 64) {
                const uint32_t i = i0 + lane;
                bool cross = false;
                uint32_t w = 0;
                uint64_t inc = 0;
                if (i < n_x && x < sub) {
                    if (packed) {
                        const uint64_t pk = d.bk_inc[at0 + i];
                        w = (uint32_t)pk & ((1u << pshift) - 1);
                        if (WIDE) w += node0;
                        inc = pk >> pshift;
                    } else {
                        w = d.bk_w[at0 + i];
                        inc = itab[w & ((1u << SEG_BITS) - 1)];
                        w = node0 + (w >> SEG_BITS);
                    }
                } else if (i < n_x) { w = s; inc = dm; }
                if (inc) {
                    const uint64_t old = atomicAdd((unsigned long long *)&target[slab + (TO_PPR && xl ? xl[w] : w)], (unsigned long long)inc);
                    if (!TO_PPR) {
                        const uint64_t thr = node_thr(t1q, d.deg[w]);
                        cross = old < thr && old + inc >= thr; // increments are positive: exactly one add crosses
                    }
                }
                if (!TO_PPR) {
                    const unsigned long long mask = __ballot(cross);
                    if (mask) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_nlist, (uint32_t)__popcll(mask));
                        base = __shfl(base, 0);
                        if (cross) s_list[base + __popcll(mask & ((1ull << lane) - 1))] = w; // <= tiny_max + 1 <= 1024 entries
                    }
                }
            }
        }
        if (TO_PPR || !d.pop_next) return;
        __syncthreads(); // every add of this bucket has returned: the crossing nodes' residues are final
        const uint32_t nl = s_nlist;
        if (!nl) return;
        if (!dk) {
            if (threadIdx.x == 0) s_gbase = atomicAdd(flc_next, nl);
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < nl; i += AT) {
                const uint32_t w = s_list[i];
                const uint64_t r = atomicExch((unsigned long long *)&d.residue[slab + w], 0ull); // algo.h:984-985
                const uint32_t pos = s_gbase + i;
                if (pos < (uint32_t)d.n) { fl_next[pos] = w; inc_next[pos] = r; }
                else atomicOr(d.err, ERR_WL_OVERFLOW);
            }
            return;
        }
        // bounded deferral: a node that only just crossed keeps its residue and is marked for the next accumulate
        // (acc[] is idle on this path: acc[0 .. 1023] collects the entries that do go to the frontier)
        if (threadIdx.x == 0) s_total = 0;
        __syncthreads();
        uint32_t ndef = 0;
        for (uint32_t i = threadIdx.x; i < nl; i += AT) {
            const uint32_t w = s_list[i];
            const uint64_t r = __atomic_load_n(&d.residue[slab + w], __ATOMIC_RELAXED);
            if ((r >> dk) < node_thr(t1q, d.deg[w])) {
                atomicOr((unsigned long long *)&dbm_out[(w - node0) >> 6], 1ull << (w & 63u));
                ndef++;
            } else {
                const uint32_t at = atomicAdd(&s_total, 1u);
                acc[at] = ((uint64_t)w << 32); // node; the residue is taken below
            }
        }
        if (ndef) atomicAdd(&s_ndef, ndef);
        __syncthreads();
        const uint32_t nf = s_total;
        if (threadIdx.x == 0) {
            if (nf) s_gbase = atomicAdd(flc_next, nf);
            if (s_ndef) { d.dflag[par ^ 1][(uint64_t)q * d.nbins + b] = 1; atomicAdd(flc_next + 2, s_ndef); }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nf; i += AT) {
            const uint32_t w = (uint32_t)(acc[i] >> 32);
            const uint64_t r = atomicExch((unsigned long long *)&d.residue[slab + w], 0ull); // algo.h:984-985
            const uint32_t pos = s_gbase + i;
            if (pos < (uint32_t)d.n) { fl_next[pos] = w; inc_next[pos] = r; }
            else atomicOr(d.err, ERR_WL_OVERFLOW);
        }
        return;
    } else {
    STAMP_DECL
    for (uint32_t i = threadIdx.x; i < BSZ; i += AT) acc[i] = 0;
    __syncthreads();
    STAMP(16);
    // messages: wave w takes sub-buckets w, w + NW, ...; ACC_UNROLL x 64 messages per iteration, all loads BRANCH-FREE
    // (clamped index, masked afterwards).  With an `if (i < cnt)` around each load hipcc puts every dependent increment
    // gather behind its own s_waitcnt vmcnt(0): serialized round trips instead of two per iteration.  (Cutting the
    // sub-buckets into 64-message units dealt round-robin to the waves -- no idle lanes but a 7-step search per unit --
    // was measured: accumulate 43 -> 59 ms per 1000 ws queries.)
#ifndef FORA_ACC_UNROLL_WIDE
#define FORA_ACC_UNROLL_WIDE 4
#endif
    constexpr int ACC_UNROLL = WIDE ? FORA_ACC_UNROLL_WIDE : 4;
    for (uint32_t x = wid; x < sub; x += NW) {
        const uint32_t n_x = s_scnt[x];
        const uint64_t at0 = bk0 + (uint64_t)x * d.bk_cap;
        for (uint32_t i0 = 0; i0 < n_x; i0 += 64 * ACC_UNROLL) { // (two sub-buckets per trip, 8 loads in flight: no gain measured)
            uint32_t mw[ACC_UNROLL];
            uint64_t mi[ACC_UNROLL], mp[ACC_UNROLL];
#pragma unroll
            for (int k = 0; k < ACC_UNROLL; k++) {
                const uint32_t i = i0 + k * 64 + lane;
                mw[k] = packed ? 0u : NT_LOAD(&d.bk_w[at0 + (i < n_x ? i : 0)]);
            }
#pragma unroll
            for (int k = 0; k < ACC_UNROLL; k++) {
                const uint32_t i = i0 + k * 64 + lane;
                mi[k] = gather ? itab[mw[k] & ((1u << SEG_BITS) - 1)] : NT_LOAD(&d.bk_inc[at0 + (i < n_x ? i : 0)]);
                mp[k] = packed ? diag::acc_load_probe(&d.bk_inc[at0 + (i < n_x ? i : 0)]) : 0ull;
            }
#pragma unroll
            for (int k = 0; k < ACC_UNROLL; k++) {
                const uint32_t i = i0 + k * 64 + lane;
                uint32_t local = mw[k] >> SEG_BITS;
                uint64_t inc = mi[k];
                if (packed) { local = (uint32_t)inc; inc >>= pshift; }
                if (i < n_x && inc) atomicAdd((unsigned long long *)&acc[local & (BSZ - 1)], (unsigned long long)inc);
                if (i < n_x && inc) diag::acc_atom_probe(&acc[local & (BSZ - 1)]);
                diag::acc_load_keep(mp[k]);
            }
        }
    }
    if (threadIdx.x == 0 && dm) atomicAdd((unsigned long long *)&acc[s & (BSZ - 1)], (unsigned long long)dm);
    if (!TO_PPR && hub_hi > hub_lo) { // the `sub` partial sums of every hub of this bin: rows of consecutive words
        const uint32_t nh = hub_hi - hub_lo;
        const uint64_t *rows = d.hubsum + (uint64_t)q * sub * d.hubs + hub_lo;
        for (uint32_t i = threadIdx.x; i < nh * sub; i += AT) {
            const uint32_t x = i / nh, h = i - x * nh;
            const uint64_t v = NT_LOAD(&rows[(uint64_t)x * d.hubs + h]);
            if (v) atomicAdd((unsigned long long *)&acc[d.hub_node[hub_lo + h] & (BSZ - 1)], (unsigned long long)v);
        }
    }
    for (uint32_t i = threadIdx.x; i < ovn; i += AT) { // increments whose bucket was full
        const uint32_t w = d.ov_w[(uint64_t)q * d.ov_cap + i];
        if ((int)(w >> BS) == b)
            atomicAdd((unsigned long long *)&acc[w & (BSZ - 1)], (unsigned long long)d.ov_inc[(uint64_t)q * d.ov_cap + i]);
    }
    __syncthreads();
    STAMP(17);
    // sweep: consecutive lanes -> consecutive nodes; all loads of a lane's 16 nodes in flight together
    uint32_t crossmask = 0;
    {
        uint64_t v[SWEEP], old[SWEEP];
        uint32_t dg[SWEEP];
#pragma unroll
        for (int k = 0; k < SWEEP; k++) v[k] = acc[k * AT + threadIdx.x];
        if (TO_PPR && xl) { // bucket order -> original ids (consecutive lanes read consecutive table entries; the adds scatter)
#pragma unroll
            for (int k = 0; k < SWEEP; k++) dg[k] = v[k] ? xl[node0 + k * AT + threadIdx.x] : 0u;
#pragma unroll
            for (int k = 0; k < SWEEP; k++) if (v[k]) old[k] = target[slab + dg[k]];
#pragma unroll
            for (int k = 0; k < SWEEP; k++) if (v[k]) target[slab + dg[k]] = old[k] + v[k]; // one bucket-order entry per node: this lane owns the word
            STAMP(18);
            STAMP_FLUSH(16);
            return;
        }
        uint32_t duemask = 0; // bit k: node k * AT + thread was deferred by the previous level
        if (din) {
#pragma unroll
            for (int k = 0; k < SWEEP; k++) { // one word per (k, wave): wave-uniform address
                const uint64_t wd = dbm_in[(k * AT + wid * 64) >> 6];
                if (wd) {
                    duemask |= (uint32_t)((wd >> lane) & 1ull) << k;
                    if (lane == 0) dbm_in[(k * AT + wid * 64) >> 6] = 0; // consumed
                }
            }
        }
        uint64_t swp = 0; // (diag::acc_sweep_probe: always 0 in the product build)
#pragma unroll
        for (int k = 0; k < SWEEP; k++) {
            old[k] = 0; dg[k] = 0;
            if (v[k] || ((duemask >> k) & 1u)) {
                const uint32_t node = node0 + k * AT + threadIdx.x;
                old[k] = target[slab + node];
                if (!TO_PPR) dg[k] = d.deg[node];
                if (!TO_PPR) swp += diag::acc_sweep_probe(&target[slab + node], &d.deg[node]);
            }
        }
        diag::acc_sweep_keep(swp);
        const bool pop = !TO_PPR && d.pop_next;
        uint32_t defmask = 0; // bit k: the node crossed but waits a level
#pragma unroll
        for (int k = 0; k < SWEEP; k++) {
            const uint32_t node = node0 + k * AT + threadIdx.x;
            const bool due = (duemask >> k) & 1u;
            if (v[k] || due) {
                const uint64_t nw = old[k] + v[k];
                bool cross = false;
                if (!TO_PPR) {
                    const uint64_t thr = node_thr(t1q, dg[k]);
                    cross = pop && (due || (old[k] < thr && nw >= thr)); // algo.h:1012
                    if (cross && !due && dk && (nw >> dk) < thr) { cross = false; defmask |= 1u << k; }
                }
                // this workgroup owns [node0, node0 + BSZ) of slot q; a crossing node gives its residue to the
                // frontier entry written below (algo.h:984-985)
                if (v[k] || cross) target[slab + node] = cross ? 0 : nw;
                if (cross) { crossmask |= 1u << k; acc[k * AT + threadIdx.x] = nw; } // own LDS word: no barrier needed
            }
        }
        if (dk) { // marks for the next accumulate: one word per (k, wave), written by its only owner
            uint32_t nd = 0;
#pragma unroll
            for (int k = 0; k < SWEEP; k++) {
                const unsigned long long m = __ballot((defmask >> k) & 1u);
                if (m) {
                    if (lane == 0) dbm_out[(k * AT + wid * 64) >> 6] = m;
                    nd += (uint32_t)__popcll(m);
                }
            }
            if (nd && lane == 0) atomicAdd(&s_ndef, nd);
        }
    }
    STAMP(18);
    if (TO_PPR) { STAMP_FLUSH(16); return; }
    // ---- rank of every crossing node in node order: node = node0 + k * AT + thread, so order by (k, wave, lane)
    constexpr int CELLS = SWEEP * NW;
    constexpr int CPL = (CELLS + 63) / 64; // cells per lane of the one wave that scans them
#pragma unroll
    for (int k = 0; k < SWEEP; k++) {
        const unsigned long long m = __ballot((crossmask >> k) & 1u);
        if (lane == 0) s_rank[k * NW + wid] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (wid == 0) { // exclusive scan of the (k, wave) counts by one wave, CPL consecutive cells per lane
        uint32_t cc[CPL], sum = 0;
#pragma unroll
        for (int j = 0; j < CPL; j++) { cc[j] = CPL * lane + j < CELLS ? s_rank[CPL * lane + j] : 0; sum += cc[j]; }
        uint32_t tot;
        uint32_t ex = wave_excl_scan(sum, tot);
#pragma unroll
        for (int j = 0; j < CPL; j++) { if (CPL * lane + j < CELLS) s_rank[CPL * lane + j] = ex; ex += cc[j]; }
        if (lane == 0) {
            s_rank[CELLS] = tot;
            s_gbase = tot ? atomicAdd(flc_next, tot) : 0u; // ONE global atomic per workgroup (the per-slot counter is a hot address)
            if (dk && s_ndef) { d.dflag[par ^ 1][(uint64_t)q * d.nbins + b] = 1; atomicAdd(flc_next + 2, s_ndef); } // (the adds to s_ndef are behind the barrier above)
        }
    }
    __syncthreads();
    STAMP(19);
    if (!s_rank[CELLS]) { STAMP_FLUSH(16); return; }
    const uint32_t gbase = s_gbase;
#pragma unroll
    for (int k = 0; k < SWEEP; k++) {
        const bool c = (crossmask >> k) & 1u;
        const unsigned long long m = __ballot(c);
        if (c) {
            const uint32_t pos = gbase + s_rank[k * NW + wid] + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (pos < (uint32_t)d.n) {
                fl_next[pos] = node0 + k * AT + threadIdx.x;
                inc_next[pos] = acc[k * AT + threadIdx.x];
            } else atomicOr(d.err, ERR_WL_OVERFLOW);
        }
    }
    STAMP(20);
    STAMP_FLUSH(16);
    }
}

// grid = (ceil(bins of the pass / G), nq) with G = Dev::acc_group bins per workgroup (narrow layout: always 1).
constexpr int ACC_GROUP_MAX = 16;
template <bool TO_PPR, bool WIDE>
__global__ void __launch_bounds__(WIDE ? ACC_THREADS_WIDE : ACC_THREADS) k_accum(Dev d, int L) {
    if (!WIDE) { accum_bin<TO_PPR, WIDE>(*dev_args(), L, (int)blockIdx.x, (int)blockIdx.y); return; }
    const bool slot_major = (d.slot_major & 2u) != 0;          // (see Dev::slot_major)
    const int q = slot_major ? blockIdx.x : blockIdx.y;
    const uint32_t bx = slot_major ? blockIdx.y : blockIdx.x;
    const uint32_t G = max(d.acc_group, 1u);
    // which of my G bins have anything to do?  Their sub-bucket counts are one contiguous stretch: a coalesced trip for all of them
    constexpr int AT = WIDE ? ACC_THREADS_WIDE : ACC_THREADS;
    constexpr int BS = WIDE ? BIN_SHIFT_WIDE : BIN_SHIFT;
    __shared__ uint32_t s_gbusy[ACC_GROUP_MAX];
    const uint32_t lb0 = bx * G;
    const uint32_t nb = min(G, (uint32_t)d.bin_cnt - lb0);
    if (G > 1) { // (uniform)
        const int par = L & 1;
        if (threadIdx.x < (uint32_t)ACC_GROUP_MAX) s_gbusy[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t sub = d.sub;
        const uint32_t *cnts = d.bk_count + ((uint64_t)q * d.pbins + lb0) * sub;
        for (uint32_t i = threadIdx.x; i < nb * sub; i += AT)
            if (cnts[i]) s_gbusy[i / sub] = 1u; // (benign race: every writer stores 1)
        if (threadIdx.x < nb) { // the other reasons a bin may have work (see accum_bin): overflow entries, hub sums, the dangling mass, deferred nodes
            const uint32_t b = (uint32_t)d.bin_lo + lb0 + threadIdx.x;
            bool busy = TEST_PATHS && !TO_PPR && d.defer_k;
            if (!TO_PPR) {
                busy = busy || d.ov_bin[par][(uint64_t)q * d.nbins + b] != 0;
                busy = busy || ((uint32_t)d.src[q] >> BS) == b; // (the dangling mass of the level, if any, lands in the source's bin)
                if (d.col_hub && d.fl_count[par][q * CSTRIDE] >= d.hub_min) busy = busy || d.hub_first[b + 1] > d.hub_first[b];
            }
            if (busy) s_gbusy[threadIdx.x] = 1u;
        }
        __syncthreads();
    }
    for (uint32_t g = 0; g < nb; g++) {
        if (G > 1 && !s_gbusy[g]) continue; // (uniform)
        accum_bin<TO_PPR, WIDE>(*dev_args(), L, (int)(lb0 + g), q); // (a fresh look at the arguments per bin: nothing rides through the loop)
        if (g + 1 < nb) __syncthreads(); // the next bin reuses the LDS of this one
    }
}

// ---- threshold rounds.  A level-synchronous push pops a node the level after it crosses its threshold, with whatever
// it has collected by then; the FIFO of algo.h:980-1017 lets it collect more before its turn, so it moves more mass
// per relaxed edge.  Rounds give that back: a slot first runs its levels against 2^(rounds-1) times the threshold
// (every pop there carries at least that much more per edge); when its frontier runs dry -- or, with round_div > 0,
// is down to 1/round_div of the round's largest frontier, so that the round's long tail of small levels is not run
// twice -- this kernel halves the threshold and sweeps the slot's residue slab once for every node at or over the
// new one: they join the next frontier, as (node, residue) entries like k_accum's.  Two rounds (2x, then 1x) cut the relaxations of the ws-sized
// headline graph from 1.27x to 1.05x of the sequential FIFO's.  Exit condition and invariants are those of algo.h:1012.
// grid = (X, nq), after the accumulate of level L.  Every workgroup of a slot takes the same decision: nothing below
// touches the frontier count before the slot's LAST workgroup is done.
#if FORA_TEST_PATHS
__global__ void __launch_bounds__(BLOCK) k_round_sweep(Dev d, int L) {
    const int q = blockIdx.y;
    const int np = (L & 1) ^ 1;
    QState *qs = &d.qs[q];
    const uint32_t ts = qs->tshift;
    if (ts == 0) return; // last round
    // the frontier the accumulate has just written: entries whose residue is already taken, the sweep appends to them.
    // Every workgroup of the slot reads the same (count, peak, shift): only the last one to finish changes them.
    const uint32_t have = d.fl_count[np][q * CSTRIDE];
    if (have != 0 && !(d.round_div > 0 && (uint64_t)have * d.round_div <= qs->peak)) return; // the round goes on
    const int lane = threadIdx.x & 63;
    const uint64_t unit = thr_unit(d.t1, ts - 1);
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t nchunk = ((uint32_t)d.n + BLOCK - 1) / BLOCK;
    for (uint32_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const uint32_t v = c * BLOCK + threadIdx.x;
        uint64_t r = 0;
        bool in = false;
        if (v < (uint32_t)d.n) {
            r = d.residue[slab + v];
            in = r && r >= node_thr(unit, d.deg[v]);
        }
        const unsigned long long mask = __ballot(in);
        if (!mask) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&d.sw_count[q * CSTRIDE], (uint32_t)__popcll(mask));
        base = __shfl(base, 0);
        if (in) {
            d.residue[slab + v] = 0; // algo.h:984-985; k_pushq_bin finishes the pop
            const uint32_t pos = have + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
            if (pos < (uint32_t)d.n) { d.fl[np][slab + pos] = v; d.inc_tab[np][(uint64_t)q * d.segq_cap + pos] = r; }
            else atomicOr(d.err, ERR_WL_OVERFLOW);
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket = atomicAdd(&d.sw_done[q * CSTRIDE], 1u);
        if (ticket == gridDim.x - 1) { // the slot's last workgroup publishes the new frontier and the new round
            __threadfence();
            const uint32_t cnt = atomicExch(&d.sw_count[q * CSTRIDE], 0u);
            d.fl_count[np][q * CSTRIDE] = min(have + cnt, (uint32_t)d.n);
            qs->tshift = ts - 1;
            qs->peak = 0;
            d.fl_count[0][q * CSTRIDE + 1] = ts - 1;
            d.fl_count[1][q * CSTRIDE + 1] = ts - 1;
            atomicExch(&d.sw_done[q * CSTRIDE], 0u);
        }
    }
}
#endif // FORA_TEST_PATHS

// ------------------------------------------------------------------ walk allocation
// One thread per (slot, node): num_s_rw and the weight r/num_s_rw, cut into <=WALK_SEG-walk
// items.  grid = (chunks, nq).
//   ALLOC_QUERY: num_s_rw = ceil(r/rsum*N)  (query.h:270,282-287; --opt: query.h:349,363-364)
//   ALLOC_TOPK : num_s_rw = ceil(r*omega), index consumed through a per-node cursor so walks
//                are never reused across rounds (query.h:558-611; online: query.h:616-632)
//   ALLOC_BOUND: top-k with bounds (query.h:639-741): no one-hop split; num_s_rw = ceil(r*omega) with the index
//                (cursor as ALLOC_TOPK), ceil(r/rsum*N) without; the round's walk total goes to round_walks[q]
enum { ALLOC_QUERY = 0, ALLOC_TOPK = 1, ALLOC_BOUND = 2 };
template <int MODE>
__global__ void __launch_bounds__(BLOCK) k_walk_alloc(Dev d, int with_idx, const uint8_t *active,
                                                      uint64_t *cursor, unsigned long long *round_walks, uint32_t epoch) {
    const bool slot_major = d.wide && (d.slot_major & 8u); // (see Dev::slot_major)
    const int q = slot_major ? blockIdx.x : blockIdx.y;
    const uint32_t bx = slot_major ? blockIdx.y : blockIdx.x, gx = slot_major ? gridDim.y : gridDim.x;
    const int lane = threadIdx.x & 63;
    if (MODE != ALLOC_QUERY && !active[q]) return;
    QState *qs = &d.qs[q];
    const uint64_t rsum_fix = FIX_ONE - qs->reserved;
    if (rsum_fix == 0) return; // query.h:267-268 / :535-536
    double check_rsum = fix2d(rsum_fix);
    uint64_t N = 0;
    if (MODE == ALLOC_QUERY) {
        if (d.opt) check_rsum *= (1 - d.alpha);       // query.h:349
        N = (uint64_t)(d.omega * check_rsum);         // query.h:270
        if (bx == 0 && threadIdx.x == 0) qs->n_rw = N;
    }
    if (MODE == ALLOC_BOUND) N = (uint64_t)(d.omega * check_rsum); // query.h:645
    const bool split = MODE == ALLOC_QUERY ? d.opt != 0 : MODE == ALLOC_TOPK ? with_idx != 0 : false;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t nchunk = ((uint32_t)d.n + BLOCK - 1) / BLOCK;
    uint64_t acc_walks = 0, acc_hit = 0;
    // SLAB_UNROLL chunks (1024 nodes) per trip, SLAB_UNROLL consecutive nodes per lane (items stay in node order: the
    // indexed walks then read rw_idx front to back), their loads issued together.  The trip's items get their place in the
    // slot's list with ONE atomic per workgroup: with one per wave a Twitter-2010-sized slab (650 k waves' worth of
    // nodes, most of them with residue) sent 650 k returning atomics to the same address -- 33 ms per launch at the
    // ~20 M/s a single address sustains.  A workgroup whose 1024 nodes hold no residue moves on at once (after a top-k
    // round well under 1 % of a slab is non-zero).
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    for (uint32_t c0 = bx * SLAB_UNROLL; c0 < nchunk; c0 += gx * SLAB_UNROLL) {
        uint64_t rr[SLAB_UNROLL];
        bool any = false;
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) {
            const uint64_t vu = (uint64_t)c0 * BLOCK + threadIdx.x * SLAB_UNROLL + u;
            rr[u] = vu < (uint64_t)d.n ? d.residue[slab + vu] : 0;
            any |= rr[u] != 0;
        }
        if (!__syncthreads_or(any)) continue; // (loading the next trip ahead of this test: no change -- the kernel's time is the per-node work of the residue nodes)
        uint64_t num[SLAB_UNROLL], incr[SLAB_UNROLL], rem[SLAB_UNROLL], iav[SLAB_UNROLL], ipos[SLAB_UNROLL];
        uint32_t nseg[SLAB_UNROLL], mine = 0;
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) {
            const uint32_t v = c0 * BLOCK + threadIdx.x * SLAB_UNROLL + u;
            uint64_t r = rr[u];
            num[u] = 0; incr[u] = 0; rem[u] = 0; iav[u] = 0; ipos[u] = 0; nseg[u] = 0;
            if (r) {
                if (split) { // query.h:363-364 / query.h:561-567
                    const uint64_t keep = mulshift62(r, d.afix);
                    d.ppr[slab + v] += keep;
                    r -= keep;
                }
                if (MODE == ALLOC_QUERY || (MODE == ALLOC_BOUND && !with_idx)) num[u] = walk_count(fix2d(r), check_rsum, N); // :727
                else num[u] = (uint64_t)ceil(fix2d(r) * d.omega); // query.h:568 / :618 / :659
                if (num[u] >> 36) { atomicOr(d.err, ERR_WIT_OVERFLOW); num[u] = 0; } // (a packed item holds walk counts below 2^36: far beyond any item list)
                if (num[u]) {
                    incr[u] = r / num[u];
                    rem[u] = r - incr[u] * num[u];
                    nseg[u] = (uint32_t)((num[u] + WALK_SEG - 1) / WALK_SEG);
                    if (with_idx) {
                        const uint64_t icnt = d.idx_cnt[v];
                        ipos[u] = d.idx_off[v];
                        if (MODE != ALLOC_QUERY) { // query.h:575-603 / :668-709
                            // a cursor word carries the batch that wrote it (epoch << 40): words of earlier batches read as 0, so
                            // the slabs are not cleared per batch (query.h:997-998; 2.3 GB per batch of 7 Twitter-2010-sized slots)
                            const uint64_t cw = cursor[slab + v];
                            const uint64_t used = (uint32_t)(cw >> 40) == epoch ? (cw & ((1ull << 40) - 1)) : 0ull;
                            iav[u] = icnt - used;
                            if (iav[u] > num[u]) iav[u] = num[u];
                            cursor[slab + v] = ((uint64_t)epoch << 40) | (used + iav[u]);
                            ipos[u] += used;
                        } else {
                            iav[u] = num[u] < icnt ? num[u] : icnt;
                        }
                        acc_hit += iav[u];
                    }
                    acc_walks += num[u];
                }
            }
            mine += nseg[u];
        }
        uint32_t tot;
        uint32_t at = block_excl_scan(mine, s_w, tot);
        if (!tot) continue; // uniform over the workgroup
        if (threadIdx.x == 0) {
            const uint32_t sb = atomicAdd(&d.wit_count[q * CSTRIDE], tot);
            if ((uint64_t)sb + tot > d.wit_cap) atomicOr(d.err, ERR_WIT_OVERFLOW);
            s_base = sb;
        }
        __syncthreads();
        const uint32_t sb = s_base;
        if ((uint64_t)sb + tot > d.wit_cap) continue;
        at += sb;
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) {
            for (uint32_t k = 0; k < nseg[u]; k++) {
                WalkItem w;
                w.j0 = (uint64_t)k * WALK_SEG;
                const uint64_t left = num[u] - w.j0;
                w.cnt = left < WALK_SEG ? (uint32_t)left : WALK_SEG;
                w.idx_pos = ipos[u] + w.j0;
                const uint64_t a = iav[u] > w.j0 ? iav[u] - w.j0 : 0;
                w.idx_n = a < w.cnt ? (uint32_t)a : w.cnt;
                w.incr = incr[u];
                w.rem = rem[u];
                w.v = c0 * BLOCK + threadIdx.x * SLAB_UNROLL + u;
                d.wit[(uint64_t)q * d.wit_cap + at++] = wit_pack(w);
            }
        }
    }
    acc_walks = wave_sum(acc_walks);
    acc_hit = wave_sum(acc_hit);
    __shared__ uint64_t s_acc[BLOCK / 64][2]; // per-slot counters: one atomic per workgroup (see k_pushq_bin)
    if (lane == 0) { s_acc[threadIdx.x >> 6][0] = acc_walks; s_acc[threadIdx.x >> 6][1] = acc_hit; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tw = 0, th = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { tw += s_acc[w][0]; th += s_acc[w][1]; }
        if (tw) atomicAdd(&qs->n_walks, (unsigned long long)tw);
        if (th) atomicAdd(&qs->n_hit, (unsigned long long)th);
        if (MODE == ALLOC_BOUND && tw) atomicAdd(&round_walks[q], (unsigned long long)tw);
    }
}

// ------------------------------------------------------------------ top-k support
// Round frontier of the incremental push (algo.h:1020-1093): every node of an active slot
// whose residue is at/over this round's threshold.  Bucketed push: a frontier entry is (node, residue taken from it),
// see k_accum.  grid = (chunks, nq).
__global__ void __launch_bounds__(BLOCK) k_topk_frontier(Dev d, const uint8_t *active) {
    // bucketed push: the workgroup collects its entries in LDS and takes list space with one atomic per ~1000 of them --
    // one per wave was tens of thousands of returning atomics per slot and round on ONE address (~20 M/s)
    constexpr uint32_t FB = 2048;
    __shared__ uint32_t s_node[FB];
    __shared__ uint64_t s_val[FB];
    __shared__ uint32_t s_cnt, s_gb;
    const int q = blockIdx.y;
    if (!active[q]) return;
    const int lane = threadIdx.x & 63;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t nchunk = ((uint32_t)d.n + BLOCK - 1) / BLOCK;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    auto flush = [&]() { // all threads; s_cnt entries -> the slot's frontier list
        const uint32_t cnt = s_cnt;
        if (threadIdx.x == 0 && cnt) s_gb = atomicAdd(&d.fl_count[0][q * CSTRIDE], cnt);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt; i += BLOCK) {
            const uint32_t pos = s_gb + i;
            if (pos < (uint32_t)d.n) { d.fl[0][slab + pos] = s_node[i]; d.inc_tab[0][(uint64_t)q * d.segq_cap + pos] = s_val[i]; }
            else atomicOr(d.err, ERR_WL_OVERFLOW);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
    };
    for (uint32_t c0 = blockIdx.x * SLAB_UNROLL; c0 < nchunk; c0 += gridDim.x * SLAB_UNROLL) { // see k_walk_alloc
        uint64_t rr[SLAB_UNROLL];
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) {
            const uint32_t vu = (c0 + u) * BLOCK + threadIdx.x;
            rr[u] = (c0 + u < nchunk && vu < (uint32_t)d.n) ? d.residue[slab + vu] : 0;
        }
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) {
            const uint32_t v = (c0 + u) * BLOCK + threadIdx.x;
            const uint64_t r = rr[u];
            if (!__ballot(r != 0)) continue;
            const bool in = r && r >= node_thr(d.t1, d.deg[v]);
            if (!d.binned) {
                wave_append(in, ((uint64_t)q << 32) | v, d.wl[0], &d.wl_count[0], d.wl_cap, d.err, ERR_WL_OVERFLOW);
                continue;
            }
            const unsigned long long mask = __ballot(in);
            if (!mask) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(mask));
            base = __shfl(base, 0);
            if (in) {
                d.residue[slab + v] = 0; // algo.h:984-985; k_pushq_bin finishes the pop
                const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
                s_node[pos] = v; s_val[pos] = r;
            }
        }
        // The flush decision must be the same in every wave (flush() holds barriers): all waves read s_cnt between two
        // barriers, where nobody adds to it.  (Round 4 read it at the top of the next trip, where a faster wave could
        // already have added: waves could disagree, pair their barriers wrongly and lose or duplicate entries.)
        __syncthreads();
        const bool full = d.binned && s_cnt > FB - SLAB_UNROLL * BLOCK;
        __syncthreads();
        if (full) flush();
    }
    if (d.binned) flush();
}

// set_graph, wide layouts: dst := src with every row padded with -1 words to whole quads (Dev::col4); one thread per node
// (a wave per 64 nodes: rows of neighbouring nodes are neighbours in both arrays).
__global__ void __launch_bounds__(BLOCK) k_pad_quads(int32_t n, const int64_t *row_ptr, const int32_t *src, const uint64_t *rowinfo4, int32_t *dst) {
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < n; v += (int64_t)gridDim.x * BLOCK) {
        const int64_t beg = row_ptr[v], deg = row_ptr[v + 1] - beg;
        int32_t *out = dst + 4 * (int64_t)(rowinfo4[v] >> 24);
        const int64_t padded = (deg + 3) & ~3ll;
        for (int64_t i = 0; i < padded; i++) out[i] = i < deg ? src[beg + i] : -1;
    }
}

// ppr := reserve for active slots (compute_ppr_with_reserve, query.h:243-253)
__global__ void __launch_bounds__(BLOCK) k_copy_slab(int32_t n, const uint64_t *src, uint64_t *dst,
                                                     const uint8_t *active) {
    const int q = blockIdx.y;
    if (active && !active[q]) return;
    const uint64_t slab = (uint64_t)q * n;
    const uint64_t step = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t v0 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v0 < (uint64_t)n; v0 += step * SLAB_UNROLL) {
        uint64_t x[SLAB_UNROLL];
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) x[u] = v0 + u * step < (uint64_t)n ? src[slab + v0 + u * step] : 0;
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) if (v0 + u * step < (uint64_t)n) dst[slab + v0 + u * step] = x[u];
    }
}

// stop test of query.h:1030: kth_ppr >= (1+eps)*delta  <=>  at least k entries >= it
__global__ void __launch_bounds__(BLOCK) k_count_above(Dev d, const uint8_t *active, double T,
                                                       unsigned long long *counts) {
    const int q = blockIdx.y;
    if (!active[q]) return;
    const uint64_t slab = (uint64_t)q * d.n;
    uint64_t acc = 0;
    const uint64_t step = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t v0 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v0 < (uint64_t)d.n; v0 += step * SLAB_UNROLL) {
        uint64_t x[SLAB_UNROLL];
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) x[u] = (v0 + u * step < (uint64_t)d.n) ? d.ppr[slab + v0 + u * step] : 0;
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) acc += x[u] && fix2d(x[u]) >= T; // T > 0
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&counts[q], (unsigned long long)acc);
}

// topk_ppr (algo.h:592-610): the k largest entries of a slot's ppr slab, score descending,
// ties by ascending id, padded with (0, 0.0).  One 1024-thread block per slot: 8-bit radix
// select of the k-th value (LDS histograms), ordered compaction, bitonic sort in LDS.
constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAXK = 1024;
// ordered compaction of a slot's non-zero ppr entries for k_topk_select: block x of slot q owns the contiguous
// node range [x*R, (x+1)*R).  grid = (X, nq), X <= 1024.
// thr != null: only entries of at least thr[q] are kept (fixed-point slabs) -- a top-k slot that stopped because k
// entries reached (1 + eps) * delta (query.h:1030) has its k largest among those, a few thousand entries instead of
// the millions of non-zero ones that one k_topk_select workgroup would read eight times over.
__global__ void __launch_bounds__(BLOCK) k_nz_count(Dev d, uint32_t R, uint32_t *counts, const double *thr) {
    __shared__ uint32_t s_w[4];
    const int q = blockIdx.y;
    const uint64_t *p = d.ppr + (uint64_t)q * d.n;
    const uint32_t lo = blockIdx.x * R, hi = min((uint32_t)d.n, lo + R);
    const double t = thr ? thr[q] : 0.0;
    uint32_t c = 0;
    for (uint32_t v0 = lo + threadIdx.x; v0 < hi; v0 += BLOCK * SLAB_UNROLL) {
        uint64_t x[SLAB_UNROLL];
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) x[u] = v0 + u * BLOCK < hi ? p[v0 + u * BLOCK] : 0;
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) c += x[u] != 0 && (!thr || fix2d(x[u]) >= t);
    }
    uint32_t tot;
    (void)block_excl_scan(c, s_w, tot);
    if (threadIdx.x == 0) counts[(uint64_t)q * gridDim.x + blockIdx.x] = tot;
}
__global__ void __launch_bounds__(BLOCK) k_nz_write(Dev d, uint32_t R, const uint32_t *counts, uint32_t *cids,
                                                    uint64_t *ckeys, uint32_t *ccount, const double *thr) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    const int q = blockIdx.y;
    const uint64_t *p = d.ppr + (uint64_t)q * d.n;
    uint32_t part = 0;
    for (uint32_t x = threadIdx.x; x < blockIdx.x; x += BLOCK) part += counts[(uint64_t)q * gridDim.x + x];
    uint32_t before;
    (void)block_excl_scan(part, s_w, before);
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base = before;
        if (blockIdx.x == gridDim.x - 1) ccount[q] = before + counts[(uint64_t)q * gridDim.x + blockIdx.x];
    }
    __syncthreads();
    const uint32_t lo = blockIdx.x * R, hi = min((uint32_t)d.n, lo + R);
    uint32_t base = s_base;
    const double t = thr ? thr[q] : 0.0;
    for (uint32_t v0 = lo; v0 < hi; v0 += BLOCK) {
        const uint32_t v = v0 + threadIdx.x;
        uint64_t x = v < hi ? p[v] : 0;
        if (thr && x && !(fix2d(x) >= t)) x = 0;
        uint32_t tot;
        __syncthreads();
        const uint32_t off = block_excl_scan(x != 0 ? 1u : 0u, s_w, tot);
        if (x) {
            cids[(uint64_t)q * d.n + base + off] = v;
            ckeys[(uint64_t)q * d.n + base + off] = x;
        }
        base += tot;
    }
}

// raw != 0: the slab holds non-negative f64 (lower bounds; their bit patterns order like the values) and the
// scores are those doubles.
// cids != null: the slot's non-zero entries were compacted in id order by k_nz_count / k_nz_write (large graphs: one
// block re-reading a 40 M-entry slab ten times costs 50 ms per 10 M nodes; the compacted list is ~1 % of it).
__global__ void __launch_bounds__(SEL_THREADS) k_topk_select(Dev d, int k, int32_t *ids, double *scores, int raw,
                                                             const uint32_t *cids, const uint64_t *ckeys, const uint32_t *ccount) {
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_key[SEL_MAXK];
    __shared__ uint32_t s_id[SEL_MAXK];
    __shared__ uint32_t s_scan[SEL_THREADS / 64];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_need, s_base_gt, s_base_eq;
    const int q = blockIdx.x;
    const uint64_t *p = cids ? ckeys + (uint64_t)q * d.n : d.ppr + (uint64_t)q * d.n;
    const uint32_t *pid = cids ? cids + (uint64_t)q * d.n : nullptr;
    const uint32_t n = cids ? ccount[q] : (uint32_t)d.n;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // ---- radix select: value of the k-th largest (0 if fewer than k positive entries)
    if (tid == 0) { s_prefix = 0; s_need = (uint32_t)k; }
    __syncthreads();
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const uint64_t prefix = s_prefix;
        const uint64_t himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
        for (uint32_t v = tid; v < n; v += SEL_THREADS) {
            const uint64_t x = p[v];
            if ((x & himask) == prefix) atomicAdd(&s_hist[(x >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t need = s_need, b = 255;
            for (;; b--) {
                const uint32_t c = s_hist[b];
                if (c >= need || b == 0) break;
                need -= c;
            }
            s_need = need; // rank wanted inside bucket b (may exceed its size only when b == 0)
            s_prefix = prefix | ((uint64_t)b << shift);
        }
        __syncthreads();
    }
    const uint64_t vk = s_prefix; // k-th largest value (0 when fewer than k positive entries)
    // ---- ordered compaction: all entries > vk, then the first `need` ids with value == vk (> 0)
    if (tid == 0) { s_base_gt = 0; s_base_eq = 0; }
    __syncthreads();
    const uint32_t need_eq = vk ? s_need : 0;
    uint32_t n_gt_total = 0;
    for (uint32_t base = 0; base < n; base += SEL_THREADS) {
        const uint32_t v = base + tid;
        const uint64_t x = v < n ? p[v] : 0;
        const bool gt = x > vk, eq = vk && x == vk;
        // block-ordered ranks of gt and eq flags
        const unsigned long long mg = __ballot(gt), me = __ballot(eq);
        const uint32_t rg = __popcll(mg & ((1ull << lane) - 1)), re = __popcll(me & ((1ull << lane) - 1));
        if (lane == 0) s_scan[wid] = (uint32_t)__popcll(mg) | ((uint32_t)__popcll(me) << 16);
        __syncthreads();
        uint32_t og = 0, oe = 0, tg = 0, te = 0;
        for (int w = 0; w < SEL_THREADS / 64; w++) {
            const uint32_t c = s_scan[w];
            if (w < wid) { og += c & 0xFFFF; oe += c >> 16; }
            tg += c & 0xFFFF; te += c >> 16;
        }
        const uint32_t bg = s_base_gt, be = s_base_eq;
        if (gt) { const uint32_t pos = bg + og + rg; if (pos < (uint32_t)SEL_MAXK) { s_key[pos] = x; s_id[pos] = pid ? pid[v] : v; } }
        __syncthreads();
        if (tid == 0) { s_base_gt = bg + tg; s_base_eq = be + te; }
        (void)re; (void)oe;
        __syncthreads();
        n_gt_total = s_base_gt;
    }
    // entries equal to vk, lowest ids first, placed after the greater ones
    if (tid == 0) s_base_eq = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n && need_eq; base += SEL_THREADS) {
        const uint32_t v = base + tid;
        const uint64_t x = v < n ? p[v] : 0;
        const bool eq = x == vk;
        const unsigned long long me = __ballot(eq);
        const uint32_t re = __popcll(me & ((1ull << lane) - 1));
        if (lane == 0) s_scan[wid] = (uint32_t)__popcll(me);
        __syncthreads();
        uint32_t oe = 0, te = 0;
        for (int w = 0; w < SEL_THREADS / 64; w++) { if (w < wid) oe += s_scan[w]; te += s_scan[w]; }
        const uint32_t be = s_base_eq;
        if (eq) {
            const uint32_t rank = be + oe + re;
            if (rank < need_eq) { const uint32_t pos = n_gt_total + rank; if (pos < (uint32_t)SEL_MAXK) { s_key[pos] = x; s_id[pos] = pid ? pid[v] : v; } }
        }
        __syncthreads();
        if (tid == 0) s_base_eq = be + te;
        __syncthreads();
        if (s_base_eq >= need_eq) break;
    }
    __syncthreads();
    uint32_t have = n_gt_total + (need_eq < s_base_eq ? need_eq : s_base_eq);
    if (have > (uint32_t)k) have = (uint32_t)k;
    // ---- bitonic sort of SEL_MAXK slots by (value desc, id asc); empty slots sort last
    for (uint32_t i = tid; i < (uint32_t)SEL_MAXK; i += SEL_THREADS)
        if (i >= have) { s_key[i] = 0; s_id[i] = 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t sz = 2; sz <= (uint32_t)SEL_MAXK; sz <<= 1) {
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < (uint32_t)SEL_MAXK; i += SEL_THREADS) {
                const uint32_t j = i ^ st;
                if (j > i) {
                    const uint64_t ki = s_key[i], kj = s_key[j];
                    const uint32_t ii = s_id[i], ij = s_id[j];
                    const bool i_first = ki > kj || (ki == kj && ii < ij); // i should come before j
                    const bool up = (i & sz) == 0;
                    if (up ? !i_first : i_first) { s_key[i] = kj; s_key[j] = ki; s_id[i] = ij; s_id[j] = ii; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < (uint32_t)k; i += SEL_THREADS) {
        const bool ok = i < have;
        ids[(uint64_t)q * k + i] = ok ? (int32_t)s_id[i] : 0;
        scores[(uint64_t)q * k + i] = ok ? (raw ? __longlong_as_double((long long)s_key[i]) : fix2d(s_key[i])) : 0.0;
    }
}

// ---- top-k with bounds (get_topk without --opt): set_ppr_bounds (algo.h:1178-1261) and if_stop (algo.h:1096-1166).
// Per node only +, *, / and sqrt of f64 in the reference's operand order (L = log(2/pfail) comes from the host), so
// the bounds equal oracle/fora_twin.c's bit for bit.  upper / lower: [slot][n] f64 (upper_bounds / lower_bounds,
// query.h:1350-1353: every key present, reset to 1 / 0 per query, :941-942).
__device__ __forceinline__ double bound_lambda(double rsum, double L, double upper_bound, double total) { // algo.h:1169-1174
    return 1.0 / 3 * L * rsum / total + sqrt(4.0 / 9.0 * L * L * rsum * rsum + 8 * total * L * rsum * upper_bound) / 2.0 / total;
}
// grid = (chunks, nq).  ppr: refined estimate of the round, reserve: the push's reserve slab.
__global__ void __launch_bounds__(BLOCK) k_bounds_update(Dev d, const uint64_t *reserve_slab, const uint8_t *active,
                                                         const unsigned long long *round_walks, double L,
                                                         double min_ppr, double sqrt_min_ppr, double *upper, double *lower) {
    const int q = blockIdx.y;
    if (!active[q]) return;
    const uint64_t rsum_fix = FIX_ONE - d.qs[q].reserved;
    if (rsum_fix == 0) return; // query.h:642-643: no walks, no bound update
    const double rsum = fix2d(rsum_fix);
    const double total = (double)round_walks[q];
    const double epsilon_v_div = sqrt(2.67 * rsum * L / total);
    const double default_epsilon_v = epsilon_v_div / sqrt_min_ppr;
    const uint64_t slab = (uint64_t)q * d.n;
    for (uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v < (uint64_t)d.n; v += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t pf = d.ppr[slab + v];
        if (!pf) continue; // algo.h:1200-1201
        const double p = fix2d(pf);
        double reserve = fix2d(reserve_slab[slab + v]);
        const double up0 = upper[slab + v], lo0 = lower[slab + v];
        double epsilon_a;
        if (up0 > reserve) epsilon_a = bound_lambda(rsum, L, up0 - reserve, total); // :1211-1215
        else epsilon_a = bound_lambda(rsum, L, 1 - reserve, total);
        const double ub_eps_a = p + epsilon_a;
        double lb_eps_a = p - epsilon_a;
        if (!(lb_eps_a > 0)) lb_eps_a = 0;
        double epsilon_v = default_epsilon_v;
        if (reserve > 0 && reserve > min_ppr) { // :1230-1233
            reserve = reserve > lo0 ? reserve : lo0;
            epsilon_v = epsilon_v_div / sqrt(reserve);
        } else if (lo0 > 0) {                   // :1235-1236
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
        if (up_bound > 0) upper[slab + v] = up_bound;
        if (low_bound >= 0) lower[slab + v] = low_bound;
    }
}
// if_stop, part 1 (algo.h:1122-1136): the k nodes with the largest lower bounds (selected by k_topk_select in raw
// mode) are marked and must satisfy upper/lower <= 1 + eps.  grid = nq, SEL_THREADS threads.
__global__ void __launch_bounds__(SEL_THREADS) k_bound_ratio(Dev d, int k, const int32_t *ids, const double *lbs,
                                                             const uint8_t *active, const double *upper, double error,
                                                             uint8_t *filter, uint32_t *fail) {
    const int q = blockIdx.x;
    if (!active[q]) return;
    const uint64_t slab = (uint64_t)q * d.n;
    bool bad = false;
    for (int i = threadIdx.x; i < k; i += SEL_THREADS) {
        const double lb = lbs[(uint64_t)q * k + i];
        if (!(lb > 0)) { bad = true; continue; } // fewer than k positive lower bounds: upper/0 = inf > error
        const uint32_t id = (uint32_t)ids[(uint64_t)q * k + i];
        filter[slab + id] = 1;
        if (upper[slab + id] / lb > error) bad = true;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(&fail[q], 1u);
}
// part 2 (algo.h:1144-1163): k-th lower bound above delta, and no unmarked node with ppr > 0 whose upper bound
// reaches it unless its own bounds are still loose.  Clears the marks.  grid = (chunks, nq).
__global__ void __launch_bounds__(BLOCK) k_bound_scan(Dev d, int k, const double *lbs, const uint8_t *active,
                                                      const double *upper, const double *lower, double delta,
                                                      double error, double loose, uint8_t *filter, uint32_t *fail) {
    const int q = blockIdx.y;
    if (!active[q]) return;
    const uint64_t slab = (uint64_t)q * d.n;
    const double low_k = lbs[(uint64_t)q * k + (k - 1)];
    bool bad = !(low_k > delta); // :1145-1147
    for (uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v < (uint64_t)d.n; v += (uint64_t)gridDim.x * BLOCK) {
        if (filter[slab + v]) { filter[slab + v] = 0; continue; }
        if (!d.ppr[slab + v]) continue;
        const double up = upper[slab + v];
        if (up > low_k * error && !(up > loose * lower[slab + v])) bad = true;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(&fail[q], 2u);
}
// upper := 1, lower := 0 for the batch (query.h:941-942)
__global__ void __launch_bounds__(BLOCK) k_bounds_reset(int32_t n, double *upper, double *lower) {
    const uint64_t slab = (uint64_t)blockIdx.y * n;
    for (uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v < (uint64_t)n; v += (uint64_t)gridDim.x * BLOCK) {
        upper[slab + v] = 1.0;
        lower[slab + v] = 0.0;
    }
}

// index sizes are computed on the host (build.h:325-334); this cuts them into items
__global__ void __launch_bounds__(BLOCK) k_index_alloc(Dev d) {
    const int lane = threadIdx.x & 63;
    const int q = 0;
    const uint32_t nchunk = ((uint32_t)d.n + BLOCK - 1) / BLOCK;
    for (uint32_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const uint32_t v = c * BLOCK + threadIdx.x;
        uint64_t num = 0, ioff = 0;
        if (v < (uint32_t)d.n) { num = d.idx_cnt[v]; ioff = d.idx_off[v]; }
        const uint32_t nseg = (uint32_t)((num + WALK_SEG - 1) / WALK_SEG);
        uint32_t tot;
        const uint32_t off = wave_excl_scan(nseg, tot);
        if (tot) {
            uint32_t sb = 0;
            if (lane == 0) sb = atomicAdd(&d.wit_count[q * CSTRIDE], tot);
            sb = __shfl(sb, 0);
            if ((uint64_t)sb + tot > d.wit_cap) {
                if (lane == 0) atomicOr(d.err, ERR_WIT_OVERFLOW);
            } else {
                for (uint32_t k = 0; k < nseg; k++) {
                    WalkItem w = {};
                    w.j0 = (uint64_t)k * WALK_SEG;
                    const uint64_t left = num - w.j0;
                    w.cnt = left < WALK_SEG ? (uint32_t)left : WALK_SEG;
                    w.idx_pos = ioff + w.j0;
                    w.v = v;
                    d.wit[sb + off + k] = wit_pack(w); // slot 0
                }
            }
        }
    }
}

// ------------------------------------------------------------------ walks
enum { WALK_TO_PPR = 0, WALK_TO_INDEX = 1 };

// One alpha-terminated walk under the Philox contract (see oracle/fora_oracle.c orc_walk;
// semantics algo.h:124-142 and, with nzh, algo.h:144-166).
__device__ __forceinline__ int32_t walk_one(const Dev &d, uint32_t start, uint64_t j, uint32_t stream,
                                            uint32_t round, int nzh, uint32_t &steps) {
    int64_t beg; uint64_t deg;
    node_row(d, start, beg, deg);
    if (deg == 0) return (int32_t)start; // algo.h:127-129
    uint32_t cur = start;
    uint32_t w[4];
    for (uint32_t t = 0;; t++) {
        if ((t & 1u) == 0)
            philox4x32_10(start, (uint32_t)j,
                          (uint32_t)((j >> 32) & 0xFFFFu) | ((round & 0xFFu) << 16) | (((t >> 1) & 0xFFu) << 24),
                          stream ^ ((t >> 9) * 0x9E3779B9u), d.seed_lo, d.seed_hi, w); // (t >> 9): steps 512.. get fresh streams
        const uint32_t ws = (t & 1u) ? w[2] : w[0], wm = (t & 1u) ? w[3] : w[1];
        if (!(nzh && t == 0) && ws < d.alpha32) return (int32_t)cur; // algo.h:131-133
        if (t) node_row(d, cur, beg, deg);
        if (deg > 0) // algo.h:134-137
            cur = (uint32_t)d.col[beg + (int64_t)(((uint64_t)wm * deg) >> 32)];
        else         // algo.h:138-140
            cur = start;
        steps++;
    }
}

// ---- wave-level staging of walk results for the bucketed accumulate ------------------------
// A wave collects (dest, weight) pairs in its own LDS area; when the area is nearly full it
// bins them by target range (LDS counters), reserves bucket space with ONE global atomic per
// (flush, bin), and stores them.  k_accum then reduces every (slot, bin) bucket in LDS.
#ifndef FORA_STAGE
#define FORA_STAGE 320
#endif
#ifndef FORA_WALK_WPE
#define FORA_WALK_WPE 6
#endif
constexpr int STAGE = FORA_STAGE; // results per wave.  Round 2 (1000 ws queries, walk kernel): 448 at 5 waves per SIMD 160.0 ms, 320 at 6 (80 VGPRs, 26 KB of LDS) 154.8,
                                  // 256 at "7" (the compiler stays at 83 VGPRs = 5) 165.0, 192 at 6 160.3; 4 waves per SIMD 180.2
struct WaveStage {
    uint64_t *pk;    // [STAGE]
    uint32_t *bcnt;  // [MAX_BINS]
    uint32_t *bbase; // [MAX_BINS]
    uint32_t *fill;  // [MAX_BINS] of the WORKGROUP: messages in its sub-bucket of every bin (see Dev::bk_w)
    uint32_t count;  // wave-uniform
    const uint32_t *xl; // not null: destinations are in WalkDG bucket order, original id = xl[dest] (only the direct-atomic fallbacks need it)
};
template <int ST>
__device__ __forceinline__ void stage_flush(const Dev &d, int q, WaveStage &st) {
    const int lane = threadIdx.x & 63;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint64_t bk0 = ((uint64_t)q * d.pbins * d.sub + blockIdx.x) * d.bk_cap; // sub-bucket blockIdx.x of bin b: + b * sub * bk_cap
    const uint64_t bstride = (uint64_t)d.sub * d.bk_cap;
    st.bcnt[lane] = 0;
    st.bcnt[lane + 64] = 0;
    __builtin_amdgcn_wave_barrier();
    constexpr uint64_t NONE = ~0ull;
    uint32_t rk[ST / 64];
    uint64_t wv[ST / 64];
#pragma unroll
    for (int k = 0; k < ST / 64; k++) {
        const uint32_t m = k * 64 + lane;
        wv[k] = NONE;
        if (m < st.count) {
            wv[k] = st.pk[m];
            rk[k] = atomicAdd(&st.bcnt[((uint32_t)wv[k] & ((1u << WPACK_SHIFT) - 1)) >> BIN_SHIFT], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    { // space in the workgroup's sub-buckets: one LDS atomic per (flush, bin) on the workgroup's fill counters; and the
      // bins' offsets inside the wave's stage
        const uint32_t c0 = lane < d.nbins ? st.bcnt[lane] : 0, c1 = lane + 64 < d.nbins ? st.bcnt[lane + 64] : 0;
        uint32_t t0, t1;
        const uint32_t o0 = wave_excl_scan(c0, t0), o1 = wave_excl_scan(c1, t1);
        if (c0) st.bbase[lane] = atomicAdd(&st.fill[lane], c0);
        if (c1) st.bbase[lane + 64] = atomicAdd(&st.fill[lane + 64], c1);
        st.bcnt[lane] = o0;           // from here on: first stage slot of the bin
        st.bcnt[lane + 64] = t0 + o1;
    }
    __builtin_amdgcn_wave_barrier();
    // sort the stage by bin in place (every entry is in registers), so that consecutive lanes store to consecutive
    // bucket slots: one write request per run instead of one per walk
#pragma unroll
    for (int k = 0; k < ST / 64; k++)
        if (wv[k] != NONE) st.pk[st.bcnt[((uint32_t)wv[k] & ((1u << WPACK_SHIFT) - 1)) >> BIN_SHIFT] + rk[k]] = wv[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < ST / 64; k++) {
        const uint32_t m = k * 64 + lane;
        if (m < st.count) {
            const uint64_t pk = st.pk[m];
            const uint32_t dd = (uint32_t)pk & ((1u << WPACK_SHIFT) - 1);
            const uint32_t b = dd >> BIN_SHIFT;
            const uint32_t pos = st.bbase[b] + (m - st.bcnt[b]);
            // non-temporal: the results are read once, by k_accum; kept out of the way of the packed targets the walk steps
            // gather through L2 (walk kernel 93.8 -> 92.2 ms per 1000 ws queries; -DFORA_STAGE_PLAIN_STORE: plain stores)
#ifndef FORA_STAGE_PLAIN_STORE
            if (pos < d.bk_cap) __builtin_nontemporal_store(pk, &d.bk_inc[bk0 + (uint64_t)b * bstride + pos]);
#else
            if (pos < d.bk_cap) d.bk_inc[bk0 + (uint64_t)b * bstride + pos] = pk;
#endif
            else atomicAdd((unsigned long long *)&d.ppr[slab + (st.xl ? st.xl[dd] : dd)], (unsigned long long)(pk >> WPACK_SHIFT)); // bucket full
        }
    }
    __builtin_amdgcn_wave_barrier();
    st.count = 0;
}
template <int ST>
__device__ __forceinline__ void stage_emit(const Dev &d, int q, WaveStage &st, bool has, uint32_t dest, uint64_t w) {
    if (has && w >= WPACK_MAXW) { // does not fit the packed word (tiny walk budgets only)
        atomicAdd((unsigned long long *)&d.ppr[(uint64_t)q * d.n + (st.xl ? st.xl[dest] : dest)], (unsigned long long)w);
        has = false;
    }
    const unsigned long long mask = __ballot(has);
    if (!mask) return;
    if (has) {
        const uint32_t pos = st.count + __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1));
        st.pk[pos] = (uint64_t)dest | (w << WPACK_SHIFT);
    }
    st.count += (uint32_t)__popcll(mask);
    if (st.count > ST - 64) stage_flush<ST>(d, q, st);
}
#define WAVE_STAGE_DECL_N(st, NWAVES, ST)                                                           \
    __shared__ uint64_t st##_pk[NWAVES][ST];                                                        \
    __shared__ uint32_t st##_bcnt[NWAVES][MAX_BINS], st##_bbase[NWAVES][MAX_BINS];                 \
    __shared__ uint32_t st##_fill[MAX_BINS];                                                        \
    WaveStage st;                                                                                   \
    st.pk = st##_pk[threadIdx.x >> 6];                                                              \
    st.bcnt = st##_bcnt[threadIdx.x >> 6]; st.bbase = st##_bbase[threadIdx.x >> 6];                 \
    st.fill = st##_fill; st.count = 0; st.xl = nullptr;
#define WAVE_STAGE_DECL(st) WAVE_STAGE_DECL_N(st, BLOCK / 64, STAGE)

// ---- indexed part of the refinement (query.h:290-296, 301-306): walks jj < idx_n of an item
// are read from rw_idx.  Streaming gather; grid = (X, nq).  With the bucketed layouts the results
// go through the same block-level chunk binning as the push (LDS histogram, one global atomic per
// (chunk, bin), bin-sorted LDS stage, run write-out) and are reduced by k_accum<true>.
template <int NB>
__global__ void __launch_bounds__(BinThreads<NB>::value) k_walk_idx(Dev d) {
    constexpr int NT = BinThreads<NB>::value; // workgroup size = walk items per tile
    constexpr int EPT = NB > MAX_BINS ? FORA_IDX_EPT_WIDE : BIN_EPT;
    constexpr uint32_t CHUNK = NT * EPT;
    constexpr uint32_t PAD = NB > MAX_BINS ? FORA_RUN_PAD_WIDE : 1;
    constexpr bool BINNED = NB > 1;
    constexpr bool WIDE = NB > MAX_BINS;
    constexpr int BS = WIDE ? BIN_SHIFT_WIDE : BIN_SHIFT; // bits of a local target: 13 narrow, 14 in the wide layouts
    constexpr uint32_t BSZ = 1u << BS;
    constexpr int IB = NT == 256 ? 8 : NT == 512 ? 9 : 10; // bits of an item index inside the tile
    constexpr int DB = WIDE ? BS : WPACK_SHIFT;     // bits of the destination in a stage word: local target / node id
    static_assert(DB + IB + 1 <= 32, "stage word: destination | item | carries-one-more-unit");
    __shared__ uint64_t s_j0[NT], s_pos[NT], s_incr[NT], s_rem[NT];
    __shared__ uint32_t s_pref[NT + 1], s_w[NT / 64];
    __shared__ uint32_t s_cnt[NB], s_lofs[NB + 1];
    __shared__ uint32_t s_fill[NB]; // results this workgroup has put into its sub-bucket of every bin (see Dev::bk_w)
    // stage: ONE word per result (destination | item << DB | extra unit << (DB + IB)) and its bin; the weight comes from the item
    __shared__ uint32_t s_msg[BINNED ? CHUNK : 1];
    __shared__ uint16_t s_bin[BINNED ? CHUNK : 1];
    const bool slot_major = NB > MAX_BINS && (d.slot_major & 4u); // (see Dev::slot_major)
    const int q = slot_major ? blockIdx.x : blockIdx.y;
    const uint32_t bx = slot_major ? blockIdx.y : blockIdx.x, gx = slot_major ? gridDim.y : gridDim.x;
    const uint32_t nitems = (uint32_t)min((uint64_t)d.wit_count[q * CSTRIDE], d.wit_cap); // (a reservation that did not fit set ERR_WIT_OVERFLOW and wrote nothing)
    if (!nitems || *d.err) return;
    const WalkItemP *items = d.wit + (uint64_t)q * d.wit_cap;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t sub = d.sub; // BINNED: == gx, this workgroup owns sub-bucket bx of every bin of the slot
    uint32_t *bkc = d.bk_count + (uint64_t)q * d.pbins * sub + bx;      // count of bin b: bkc[b * sub]
    const uint64_t bk0 = ((uint64_t)q * d.pbins * sub + bx) * d.bk_cap; // sub-bucket of bin b: bk0 + b * sub * bk_cap
    const uint32_t bin_lo = (uint32_t)d.bin_lo, bin_cnt = (uint32_t)d.bin_cnt;
    if (BINNED) for (uint32_t i = threadIdx.x; i < (uint32_t)NB; i += NT) {
        s_cnt[i] = 0;
        s_fill[i] = i < bin_cnt ? bkc[(uint64_t)i * sub] : 0;
    }
    constexpr uint32_t GRAN = FORA_TILE_GRAN_WALK; // see tile_pos
    const uint32_t ntiles = (nitems + NT - 1) / NT;
    const uint32_t seg_len = ntiles * GRAN;
    for (uint32_t tile = bx; tile < ntiles; tile += gx) {
        const uint32_t i = tile_pos<GRAN>(threadIdx.x, tile, seg_len);
        uint32_t cnt = 0;
        if (i < nitems) {
            const WalkItem w = wit_load(&items[i]);
            s_j0[threadIdx.x] = w.j0; s_pos[threadIdx.x] = w.idx_pos;
            s_incr[threadIdx.x] = w.incr; s_rem[threadIdx.x] = w.rem;
            cnt = w.idx_n;
        }
        uint32_t total;
        const uint32_t pre = block_excl_scan_n<NT>(cnt, s_w, total);
        s_pref[threadIdx.x] = pre;
        if (threadIdx.x == 0) s_pref[NT] = total;
        __syncthreads();
        for (uint32_t cb = 0; cb < total; cb += CHUNK) {
            uint32_t dest[EPT], rank[EPT], li[EPT];
            const uint32_t e0 = cb + threadIdx.x * EPT;
            uint32_t lo = 0;
            if (e0 < total) {
                uint32_t hi = NT;
#pragma unroll
                for (int it = 0; it < IB; it++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_pref[mid] <= e0) lo = mid; else hi = mid;
                }
            }
#pragma unroll
            for (int k = 0; k < EPT; k++) { // items can be empty here (idx_n == 0): advance past them
                const uint32_t e = e0 + k;
                if (e < total) while (s_pref[lo + 1] <= e) lo++;
                li[k] = lo;
            }
#pragma unroll
            for (int k = 0; k < EPT; k++) { // straight-line: all EPT gathers in flight together
                const uint32_t e = e0 + k;
                dest[k] = 0xFFFFFFFFu;
                if (e < total) {
                    const uint32_t jj = e - s_pref[li[k]];
                    dest[k] = (uint32_t)d.rw_idx[s_pos[li[k]] + jj];                       // query.h:292
                    li[k] |= (s_j0[li[k]] + jj < s_rem[li[k]] ? 1u : 0u) << IB;              // query.h:293: the first `rem` walks carry one more unit
                }
            }
            if (!BINNED) {
#pragma unroll
                for (int k = 0; k < EPT; k++)
                    if (dest[k] != 0xFFFFFFFFu)
                        atomicAdd((unsigned long long *)&d.ppr[slab + dest[k]],
                                  (unsigned long long)(s_incr[li[k] & (NT - 1)] + (li[k] >> IB)));
                continue;
            }
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                if (dest[k] != 0xFFFFFFFFu && (dest[k] >> BS) - bin_lo >= bin_cnt) dest[k] = 0xFFFFFFFFu; // another pass
                if (dest[k] != 0xFFFFFFFFu) rank[k] = atomicAdd(&s_cnt[(dest[k] >> BS) - bin_lo], 1u);
            }
            __syncthreads();
            uint32_t staged; // results of this chunk that belong to the pass's bins
            {
                constexpr int PER = (NB + NT - 1) / NT;
                uint32_t c[PER], mine = 0;
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const uint32_t b = threadIdx.x * PER + j;
                    c[j] = b < (uint32_t)d.bin_cnt && b < (uint32_t)NB ? s_cnt[b] : 0;
                    mine += c[j];
                }
                uint32_t ctot;
                uint32_t pre2 = block_excl_scan_n<NT>(mine, s_w, ctot);
                staged = ctot;
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const uint32_t b = threadIdx.x * PER + j;
                    if (b < (uint32_t)NB) {
                        s_lofs[b] = pre2;
                        pre2 += c[j];
                        if (c[j]) { // space in the workgroup's own sub-bucket: a counter in LDS, no global atomic
                            s_fill[b] += (c[j] + (PAD - 1)) & ~(uint32_t)(PAD - 1);
                            s_cnt[b] = 0;
                        }
                    }
                }
                if (threadIdx.x == NT - 1) s_lofs[NB] = ctot;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                if (dest[k] != 0xFFFFFFFFu) {
                    const uint32_t b = (dest[k] >> BS) - bin_lo;
                    const uint32_t sp = s_lofs[b] + rank[k];
                    s_msg[sp] = (WIDE ? dest[k] & (BSZ - 1) : dest[k]) | (li[k] << DB);
                    s_bin[sp] = (uint16_t)b;
                }
            }
            __syncthreads();
            for (uint32_t m = threadIdx.x; m < staged; m += NT) { // consecutive lanes -> consecutive sub-bucket slots
                const uint32_t e = s_msg[m], b = s_bin[m];
                const uint32_t dd = e & ((1u << DB) - 1);
                const uint64_t wgt = s_incr[(e >> DB) & (uint32_t)(NT - 1)] + (e >> (DB + IB));
                const uint32_t crun = s_lofs[b + 1] - s_lofs[b];
                const uint32_t pos = s_fill[b] - ((crun + (PAD - 1)) & ~(uint32_t)(PAD - 1)) + (m - s_lofs[b]); // s_fill already counts this chunk's (padded) run
                const bool fits = wgt < (WIDE ? WIDE_MAXV : WPACK_MAXW);
                if (pos < d.bk_cap) d.bk_inc[bk0 + (uint64_t)b * sub * d.bk_cap + pos] = fits ? (uint64_t)dd | (wgt << DB) : 0ull;
                if (pos >= d.bk_cap || !fits) // sub-bucket full / weight too large for the packed word: direct atomic, same sum
                    atomicAdd((unsigned long long *)&d.ppr[slab + (WIDE ? ((bin_lo + b) << BS) | dd : dd)], (unsigned long long)wgt);
            }
            if (PAD > 1) { // null words up to the sector boundary (see k_pushq_bin)
                for (uint32_t b = threadIdx.x; b < bin_cnt; b += NT) {
                    const uint32_t crun = s_lofs[b + 1] - s_lofs[b];
                    const uint32_t prun = (crun + (PAD - 1)) & ~(uint32_t)(PAD - 1);
                    for (uint32_t i = crun; i < prun; i++) {
                        const uint32_t pos = s_fill[b] - prun + i;
                        if (pos < d.bk_cap) d.bk_inc[bk0 + (uint64_t)b * sub * d.bk_cap + pos] = 0ull;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (BINNED) for (uint32_t i = threadIdx.x; i < bin_cnt; i += NT) bkc[(uint64_t)i * sub] = s_fill[i];
}

// ---- online walks (query.h:297-300, 320-323; build.h:344-354).  grid = (X, nq).
// A block stages 256 items of one slot in LDS; each of its 4 waves owns a contiguous quarter
// of the tile's walks and hands them out dynamically: every second iteration idle lanes
// (ballot + prefix popcount) take the next walk numbers of the wave's range, so no lane waits
// for the longest walk of the wave.  Walks start on even iterations only, hence all running
// walks of a wave share step parity and the Philox call (one per two steps) is wave-uniform.
template <int MODE>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FORA_WALK_WPE, 8))) k_walk_online(Dev d, uint32_t round, int nzh, int32_t *idx_out) {
    __shared__ uint64_t s_j0[BLOCK], s_pos[BLOCK], s_incr[BLOCK], s_rem[BLOCK];
    __shared__ uint32_t s_v[BLOCK], s_idxn[BLOCK], s_pref[BLOCK + 1], s_w[4];
    const int q = blockIdx.y;
    const uint32_t nitems = (uint32_t)min((uint64_t)d.wit_count[q * CSTRIDE], d.wit_cap); // (see k_walk_idx)
    if (!nitems || *d.err) return;
    const WalkItemP *items = d.wit + (uint64_t)q * d.wit_cap;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t stream = MODE == WALK_TO_INDEX ? 0xFFFFFFFFu : (uint32_t)d.src[q];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t steps = 0;
    WAVE_STAGE_DECL(st)
    const bool staged = MODE == WALK_TO_PPR && d.binned && !d.wide; // then gridDim.x == d.sub: workgroup x fills sub-bucket x
    uint32_t *bkc = d.bk_count + (uint64_t)q * d.pbins * d.sub + blockIdx.x; // count of bin b: bkc[b * sub]
    if (staged) {
        for (uint32_t i = threadIdx.x; i < (uint32_t)MAX_BINS; i += BLOCK) st.fill[i] = i < (uint32_t)d.nbins ? bkc[(uint64_t)i * d.sub] : 0;
        __syncthreads();
    }
    for (uint32_t tbase = blockIdx.x * BLOCK; tbase < nitems; tbase += gridDim.x * BLOCK) {
        const uint32_t i = tbase + threadIdx.x;
        uint32_t cnt = 0;
        if (i < nitems) {
            const WalkItem w = wit_load(&items[i]);
            s_j0[threadIdx.x] = w.j0; s_pos[threadIdx.x] = w.idx_pos;
            s_incr[threadIdx.x] = w.incr; s_rem[threadIdx.x] = w.rem;
            s_v[threadIdx.x] = w.v;
            const uint32_t in_idx = MODE == WALK_TO_INDEX ? 0u : w.idx_n;
            s_idxn[threadIdx.x] = in_idx;
            cnt = w.cnt - in_idx;
        }
        uint32_t total;
        const uint32_t pre = block_excl_scan(cnt, s_w, total);
        s_pref[threadIdx.x] = pre;
        if (threadIdx.x == 0) s_pref[BLOCK] = total;
        __syncthreads();
        const uint32_t wend = (uint32_t)(((uint64_t)total * (wid + 1)) >> 2);
        uint32_t wptr = (uint32_t)(((uint64_t)total * wid) >> 2); // next unassigned walk of this wave
        uint32_t cur_item = 0;                                       // item holding walk wptr (wave-uniform)
        {
            uint32_t hi = BLOCK;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const uint32_t mid = (cur_item + hi) >> 1;
                if (s_pref[mid] <= wptr) cur_item = mid; else hi = mid;
            }
        }
        bool active = false;
        uint32_t cur = 0, start = 0, t = 0;
        uint64_t wj = 0, wgt = 0, opos = 0, deg0 = 0;
        int64_t beg0 = 0;
        uint32_t rw[4] = {0, 0, 0, 0};
        for (uint32_t it = 0;; it++) {
            int32_t done = -1; // endpoint to emit this iteration
            if ((it & 1u) == 0) {
                const unsigned long long idle = __ballot(!active);
                const uint32_t avail = wend - wptr;
                if (!avail && idle == ~0ull) break;
                if (avail && idle) {
                    const uint32_t rank = __popcll(idle & ((1ull << lane) - 1));
                    if (!active && rank < avail) {
                        const uint32_t e = wptr + rank;
                        uint32_t item = cur_item;
                        while (s_pref[item + 1] <= e) item++;
                        const uint32_t jj = s_idxn[item] + (e - s_pref[item]); // online walks follow the indexed ones
                        wj = s_j0[item] + jj;
                        start = s_v[item];
                        if (MODE == WALK_TO_INDEX) opos = s_pos[item] + jj;
                        else wgt = s_incr[item] + (wj < s_rem[item] ? 1 : 0);
                        if (d.rp32) { // keep the whole walk inside the compact copy's working set
                            const U32Pair rp = *(const U32Pair *)(d.rp32 + start);
                            beg0 = rp.a;
                            deg0 = rp.b - rp.a;
                        } else node_row(d, start, beg0, deg0);
                        cur = start;
                        t = 0;
                        if (deg0 == 0) done = (int32_t)start; // algo.h:127-129
                        else active = true;
                    }
                    const uint32_t want = (uint32_t)__popcll(idle);
                    wptr += want < avail ? want : avail;
                    while (wptr < wend && s_pref[cur_item + 1] <= wptr) cur_item++;
                }
                if (active)
                    philox4x32_10(start, (uint32_t)wj,
                                  (uint32_t)((wj >> 32) & 0xFFFFu) | ((round & 0xFFu) << 16) | (((t >> 1) & 0xFFu) << 24),
                                  stream ^ ((t >> 9) * 0x9E3779B9u), d.seed_lo, d.seed_hi, rw);
            }
            if (active) {
                const uint32_t ws = (it & 1u) ? rw[2] : rw[0], wm = (it & 1u) ? rw[3] : rw[1];
                if (!(nzh && t == 0) && ws < d.alpha32) { // algo.h:131-133
                    done = (int32_t)cur;
                    active = false;
                } else {
                    if (t) cur = walk_move(d, cur, start, wm);                                             // algo.h:134-140
                    else if (d.rp32) cur = colp_at(d, (uint64_t)beg0 + (((uint64_t)wm * deg0) >> 32));    // deg0 > 0 here
                    else cur = (uint32_t)d.col[beg0 + (int64_t)(((uint64_t)wm * deg0) >> 32)];
                    t++;
                    steps++;
                }
            }
            if (staged) {
                stage_emit<STAGE>(d, q, st, done >= 0, (uint32_t)done, wgt);                      // query.h:299,322
            } else if (done >= 0) {
                if (MODE == WALK_TO_INDEX) idx_out[opos] = done;                                  // build.h:346-353
                else atomicAdd((unsigned long long *)&d.ppr[slab + (uint32_t)done], (unsigned long long)wgt);
            }
        }
        __syncthreads();
    }
    if (staged && st.count) stage_flush<STAGE>(d, q, st);
    if (staged) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (uint32_t)d.nbins; i += BLOCK) bkc[(uint64_t)i * d.sub] = st.fill[i];
    }
    // one atomic per workgroup: every wave of every slot adds to the same word, and one address takes ~20 M atomics/s
    const uint64_t ws = wave_sum((uint64_t)steps);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)ws; // < 2^32 steps per wave
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long bs = (unsigned long long)s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (bs) atomicAdd(d.tot_steps, bs);
    }
}

// ---- online walks over the degree-grouped copy (WalkDG; narrow layout, results staged per wave).  grid = (sub, nq),
// DG_THREADS threads.  Same walks as k_walk_online bit for bit: the Philox counter keeps the ORIGINAL id of the start
// node, rows keep file order, so walk j of a node takes the same edges and ends at the same node.  What changes is the
// cost of a step: the out-degree and first edge of the current node come from its copy id and two LDS reads (block ->
// class, class record), so a step is ONE divergent gather (the packed target) instead of two dependent ones, and the
// 1.1 MB of row offsets leave the gathered working set (ws-sized: 6.6 -> 5.5 MB against 4 MB of L2 per XCD).  An
// iteration runs both steps of a Philox call; the endpoint's original id (one gather per WALK) is loaded at the end
// of the iteration and consumed by the next one's emission, so its latency hides behind the refill and the Philox
// rounds.  A workgroup is 8 waves sharing one copy of the tables.
#ifndef FORA_DG_THREADS
#define FORA_DG_THREADS 512
#endif
#ifndef FORA_DG_STAGE
#define FORA_DG_STAGE 256
#endif
#ifndef FORA_DG_WPE
#define FORA_DG_WPE 6
#endif
constexpr int DG_THREADS = FORA_DG_THREADS;
constexpr int DG_STAGE = FORA_DG_STAGE;
#ifndef FORA_DG_TILE
#define FORA_DG_TILE 256
#endif
constexpr int DG_TILE = FORA_DG_TILE; // walk items per tile (at most DG_THREADS)
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
template <bool BITS32>
__device__ __forceinline__ uint32_t dg_colp_at(const WalkDG &g, uint32_t e) {
    if (BITS32) { // narrow layout: entries of at most 20 bits, so ONE byte-aligned dword holds an entry (bit offset & 7 plus its width <= 32)
        const uint32_t at = g.bits * e;
        const uint32_t w = *(const u32_unaligned *)((const char *)g.colp + (at >> 3));
        return (w >> (at & 7)) & ((1u << g.bits) - 1u);
    }
    uint32_t word, sh;
    if (BITS32) { const uint32_t at = g.bits * e; word = at >> 5; sh = at & 31; }
    else { const uint64_t at = (uint64_t)g.bits * e; word = (uint32_t)(at >> 5); sh = (uint32_t)at & 31; }
    const U32Pair cw = *(const U32Pair *)(g.colp + word);
    const uint64_t both = ((uint64_t)cw.b << 32) | cw.a;
    return (uint32_t)(both >> sh) & ((1u << g.bits) - 1u);
}
template <bool NZH, bool BITS32, bool XL>
__global__ void __launch_bounds__(DG_THREADS) __attribute__((amdgpu_waves_per_eu(FORA_DG_WPE, 8))) k_walk_dg(Dev d, uint32_t round) {
    constexpr int NW = DG_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) uint64_t dg_lds64[]; // (16-byte aligned: the records below are read as uint4) // XL: hub accumulators [H] (u64) | first[nrec] | deg[nrec] | base[nrec] | T[nblk] (bytes)
    // Walk items are staged per WAVE, WT at a time: a walk that has started lives in its lane's registers, so the wave
    // loads its next WT items as soon as the walks of the current ones are handed out -- lanes never wait for the longest
    // walk of a tile to end, and the loop has no workgroup barrier.  (First form: 256 items per workgroup between two
    // barriers, an eighth of their walks per wave: a tile's last walks ran on a few lanes for ~9 iterations per ~35.)
    constexpr int WT = DG_TILE / NW;
    static_assert(WT >= 1 && WT <= 64, "a lane loads one item of its wave's tile");
    __shared__ uint64_t w_j0[NW][WT], w_incr[NW][WT], w_rem[NW][WT];
    __shared__ uint32_t w_v[NW][WT], w_vp[NW][WT], w_idxn[NW][WT], w_pref[NW][WT + 1], s_w[NW];
    // A fifth of all walks stop where they started (algo.h:131-133 at the first step): with XL their weights are summed per
    // item in LDS (w_self) and leave as ONE result per item when the wave replaces its tile -- a walk that outlives its tile
    // (its tag names the tile it came from) is emitted on its own as before.
    __shared__ unsigned long long w_self[XL ? NW : 1][XL ? WT : 1];
    const int q = blockIdx.y;
    const uint32_t nitems = (uint32_t)min((uint64_t)d.wit_count[q * CSTRIDE], d.wit_cap); // (see k_walk_idx)
    if (!nitems || *d.err) return;
    const WalkDG &g = d.dg;
    const uint32_t H = g.H, ts = g.ts, nblk1 = g.nblk ? g.nblk - 1 : 0;
    unsigned long long *s_hub = (unsigned long long *)dg_lds64;
    uint32_t *dg_lds = (uint32_t *)(dg_lds64 + (XL ? ((H + 1) & ~1u) : 0)); // (whole 16-byte words of hub accumulators)
    // a record in LDS: (first copy id, out-degree, first edge, -) as ONE 16-byte word -- a move reads it with one ds_read_b128
    // (rounds 3-4: three arrays, three reads and their addresses per step)
    uint4 *s_rec = (uint4 *)dg_lds;                       // [nrec]; 16-byte aligned: dg_lds64 is, H * 8 keeps it
    const uint8_t *s_T = (const uint8_t *)(dg_lds + 4 * g.nrec);
    for (uint32_t i = threadIdx.x; i < g.nrec; i += DG_THREADS) s_rec[i] = make_uint4(g.rec[i], g.rec[g.nrec + i], g.rec[2 * g.nrec + i], 0u);
    for (uint32_t i = threadIdx.x; i < (g.nblk + 3) / 4; i += DG_THREADS) dg_lds[4 * g.nrec + i] = ((const uint32_t *)g.T)[i];
    if (XL) for (uint32_t i = threadIdx.x; i < H; i += DG_THREADS) s_hub[i] = 0;
    const WalkItemP *items = d.wit + (uint64_t)q * d.wit_cap;
    const uint64_t slab = (uint64_t)q * d.n;
    const uint32_t stream = (uint32_t)d.src[q];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t steps = 0;
    WAVE_STAGE_DECL_N(st, NW, DG_STAGE)
    if (XL) st.xl = g.invb;
    uint32_t *bkc = d.bk_count + (uint64_t)q * d.pbins * d.sub + blockIdx.x; // count of bin b: bkc[b * sub]
    for (uint32_t i = threadIdx.x; i < (uint32_t)MAX_BINS; i += DG_THREADS) st.fill[i] = i < (uint32_t)d.nbins ? bkc[(uint64_t)i * d.sub] : 0;
    __syncthreads();
    // one move (algo.h:134-140) from copy id `cur` with random word wm
    auto move = [&](uint32_t cur, uint32_t startp, uint32_t wm) -> uint32_t {
        const uint32_t tb = s_T[min((cur - H) >> ts, nblk1)]; // (a hub: cur - H wraps, the index is clamped, the byte unused -- no branch around the read)
        const uint32_t r = cur < H ? cur : H + tb;
        const uint4 rc = s_rec[r];
        const uint32_t dg = rc.y;
        const uint32_t e = rc.z + (cur - rc.x) * dg + __umulhi(wm, dg);
        const uint32_t nx = dg_colp_at<BITS32>(g, dg ? e : 0u);
        return dg ? nx : startp;
    };
    // !XL: result waiting for its original id (loaded at the end of the previous iteration)
    bool pend = false;
    uint32_t pend_node = 0;
    uint64_t pend_w = 0;
    unsigned long long *s_self = w_self[XL ? wid : 0];
    if (XL && lane < WT) s_self[lane] = 0;
    uint32_t gen = 0, tag = 0; // tiles this wave has loaded; (tile << 6 | item) of this lane's walk
    // the tile's self sums -> results like any other endpoint (hub accumulators / the wave's stage), then zero
    auto flush_self = [&]() {
        if (!XL) return;
        const unsigned long long sv = lane < WT ? s_self[lane] : 0ull;
        const uint32_t dn = lane < WT ? w_vp[wid][lane] : 0u;
        if (lane < WT && sv) s_self[lane] = 0;
        if (sv && dn < H) atomicAdd(&s_hub[dn], sv); // LDS
        const uint32_t u = dn - H, blk = u >> 6;
        const uint32_t qd = __umulhi(blk, g.nbx_magic);
        const uint32_t dest = ((blk - qd * g.nbx) << BIN_SHIFT) | (qd << 6) | (u & 63u);
        stage_emit<DG_STAGE>(d, q, st, sv != 0 && dn >= H, dest, (uint64_t)sv);
    };
    uint64_t *s_j0 = w_j0[wid], *s_incr = w_incr[wid], *s_rem = w_rem[wid];
    uint32_t *s_v = w_v[wid], *s_vp = w_vp[wid], *s_idxn = w_idxn[wid], *s_pref = w_pref[wid];
    const uint32_t ntiles = (nitems + WT - 1) / WT;
    uint32_t tile = blockIdx.x * NW + wid;      // the wave's next tile of WT items
    const uint32_t tstride = gridDim.x * NW;
    uint32_t wptr = 0, wend = 0;                // walks of the current tile: wptr .. wend - 1 are not handed out yet (wave-uniform)
    uint32_t cur_item = 0;                      // item holding walk wptr (wave-uniform)
    bool active = false;
    uint32_t cur = 0, start = 0, startp = 0, t = 0;
    uint64_t wj = 0, wgt = 0;
    for (;;) {
        int32_t done = -1; // endpoint (copy id) reached this iteration
        const unsigned long long idle = __ballot(!active);
        if (idle && wptr == wend) { // lanes are free and the tile is handed out: the next tile that has walks
            while (wptr == wend && tile < ntiles) {
                if (gen) flush_self();
                gen++;
                const uint32_t i = tile * WT + lane;
                uint32_t cnt = 0;
                if (lane < WT && i < nitems) {
#ifndef FORA_DG_PLAIN_ITEMS
                    WalkItem w; // read once: keep the items out of the way of the packed targets in L2
                    {
                        const uint64_t *wp = (const uint64_t *)&items[i];
                        const uint64_t x0 = NT_LOAD(wp), x1 = NT_LOAD(wp + 1), x3 = NT_LOAD(wp + 3); // (online walks: no index position)
                        w = wit_unpack(x0, x1, 0, x3);
                    }
#else
                    const WalkItem w = wit_load(&items[i]);
#endif
                    s_j0[lane] = w.j0; s_incr[lane] = w.incr; s_rem[lane] = w.rem;
                    s_v[lane] = w.v;
                    s_vp[lane] = g.perm[w.v];
                    s_idxn[lane] = w.idx_n;
                    cnt = w.cnt - w.idx_n;
                }
                uint32_t total;
                const uint32_t pre = wave_excl_scan(cnt, total);
                if (lane < WT) s_pref[lane] = pre;
                if (lane == 0) s_pref[WT] = total;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the wave's own LDS writes, read by its lanes below)
                __builtin_amdgcn_wave_barrier();
                wptr = 0; wend = total; cur_item = 0;
                tile += tstride;
                while (wptr < wend && (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pref[cur_item + 1]) <= wptr) cur_item++; // items without online walks
            }
        }
        const uint32_t avail = wend - wptr;
        if (!avail && idle == ~0ull) break; // (no tile left either)
        if (avail && idle) {
            const uint32_t rank = __popcll(idle & ((1ull << lane) - 1));
            if (!active && rank < avail) {
                const uint32_t e = wptr + rank;
                uint32_t item = cur_item;
                while (s_pref[item + 1] <= e) item++;
                const uint32_t jj = s_idxn[item] + (e - s_pref[item]); // online walks follow the indexed ones
                wj = s_j0[item] + jj;
                start = s_v[item];
                startp = s_vp[item];
                wgt = s_incr[item] + (wj < s_rem[item] ? 1 : 0); // (!XL: the lane's previous result already waits in pend_w)
                cur = startp;
                t = 0;
                tag = (gen << 6) | item;
                if (startp >= g.zero_first) done = (int32_t)startp; // algo.h:127-129
                else active = true;
            }
            const uint32_t want = (uint32_t)__popcll(idle);
            wptr += want < avail ? want : avail;
            while (wptr < wend && (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pref[cur_item + 1]) <= wptr) cur_item++;
            __builtin_amdgcn_wave_barrier(); // (every lane has read its item before the tile can be replaced)
        }
        if (active) {
            uint32_t rw[4];
            philox4x32_10(start, (uint32_t)wj,
                          (uint32_t)((wj >> 32) & 0xFFFFu) | ((round & 0xFFu) << 16) | (((t >> 1) & 0xFFu) << 24),
                          stream ^ ((t >> 9) * 0x9E3779B9u), d.seed_lo, d.seed_hi, rw);
            if (!(NZH && t == 0) && rw[0] < d.alpha32) { // algo.h:131-133
                done = (int32_t)cur;
                active = false;
            } else {
                cur = move(diag::dg_from(cur, t), startp, rw[1]); // (diag::dg_from: the identity in the product build)
                steps++;
                if (rw[2] < d.alpha32) {
                    t++;
                    done = (int32_t)cur;
                    active = false;
                } else {
                    cur = move(diag::dg_from(cur, 1u), startp, rw[3]);
                    t += 2;
                    steps++;
                }
            }
        }
        if (XL) { // query.h:299,322
            bool ended = done >= 0;
            const uint32_t dn = (uint32_t)done;
            if (ended && dn == startp && (tag >> 6) == gen) { // back where it started, and the item is still in the tile
                atomicAdd(&s_self[tag & 63u], (unsigned long long)wgt); // LDS
                ended = false;
            }
            if (ended && dn < H) atomicAdd(&s_hub[dn], (unsigned long long)wgt); // LDS
            const uint32_t u = dn - H, blk = u >> 6;
            const uint32_t qd = __umulhi(blk, g.nbx_magic);                      // blk / nbx
            const uint32_t dest = ((blk - qd * g.nbx) << BIN_SHIFT) | (qd << 6) | (u & 63u);
            stage_emit<DG_STAGE>(d, q, st, ended && dn >= H, dest, wgt);
        } else {
            stage_emit<DG_STAGE>(d, q, st, pend, pend_node, pend_w);
            pend = done >= 0;
            if (pend) { pend_node = g.inv[done]; pend_w = wgt; }
        }
    }
    if (!XL) stage_emit<DG_STAGE>(d, q, st, pend, pend_node, pend_w);
    if (gen) flush_self();
    if (st.count) stage_flush<DG_STAGE>(d, q, st);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)d.nbins; i += DG_THREADS) bkc[(uint64_t)i * d.sub] = st.fill[i];
    if (XL) for (uint32_t i = threadIdx.x; i < H; i += DG_THREADS) { // the hubs' share of this workgroup's walks
        const unsigned long long hv = s_hub[i];
        if (hv) atomicAdd((unsigned long long *)&d.ppr[slab + g.inv[i]], hv);
    }
    const uint64_t ws = wave_sum((uint64_t)steps);
    if (lane == 0) s_w[wid] = (uint32_t)ws; // < 2^32 steps per wave
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bs = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) bs += s_w[w];
        if (bs) atomicAdd(d.tot_steps, bs);
    }
}

// ------------------------------------------------------------------ epilogue / hooks
// sum of each slot's ppr slab (mass check); grid = (chunks, nq)
__global__ void __launch_bounds__(BLOCK) k_ppr_sum(Dev d) {
    const int q = blockIdx.y;
    const uint64_t slab = (uint64_t)q * d.n;
    uint64_t acc = 0;
    const uint64_t step = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t v0 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v0 < (uint64_t)d.n; v0 += step * SLAB_UNROLL) {
        uint64_t x[SLAB_UNROLL];
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) x[u] = v0 + u * step < (uint64_t)d.n ? d.ppr[slab + v0 + u * step] : 0;
#pragma unroll
        for (int u = 0; u < SLAB_UNROLL; u++) acc += x[u];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&d.qs[q].ppr_sum, (unsigned long long)acc);
}

// stage hook: walk allocation on caller-supplied f64 residues (reference arithmetic)
__global__ void __launch_bounds__(BLOCK) k_walk_counts_f64(int32_t n, const double *residue, double rsum,
                                                           double omega, double alpha, int opt,
                                                           uint64_t *num_s_rw, uint64_t *n_rw) {
    double check_rsum = rsum;
    if (opt) check_rsum *= (1 - alpha);
    const uint64_t N = check_rsum == 0.0 ? 0 : (uint64_t)(omega * check_rsum);
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v == 0) *n_rw = N;
    if (v >= n) return;
    double r = residue[v];
    uint64_t num = 0;
    if (r > 0 && check_rsum != 0.0) {
        if (opt) r = r * (1 - alpha);
        num = walk_count(r, check_rsum, N);
    }
    num_s_rw[v] = num;
}

// stage hook: raw walk endpoints
__global__ void __launch_bounds__(BLOCK) k_walks_raw(Dev d, uint32_t stream, uint32_t round, int nzh,
                                                     const int32_t *starts, const uint64_t *js, int64_t count,
                                                     int32_t *dests) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    uint32_t steps = 0;
    dests[i] = walk_one(d, (uint32_t)starts[i], js[i], stream, round, nzh, steps);
}

} // namespace fora
