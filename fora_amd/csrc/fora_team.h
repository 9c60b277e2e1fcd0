// fora_team.h -- the "team push": forward push (algo.h:954-1018) with the residue of a slot resident in LDS for the
// whole push.
//
// The bucketed push of fora_kernels.h (k_pushq_bin + k_accum) keeps one level of increments in LDS and sends the
// residue HBM -> LDS sum -> HBM every level: at the mid-size levels of a ws-sized query that rewrite of the slot's slab
// is most of the bytes the pair moves.  Here a TEAM of T workgroups, one per CU, owns a slot from its
// first level to (nearly) its last: member c keeps the residue of the nodes it owns in LDS (64-node blocks dealt
// round-robin: owner(v) = (v >> 6) % T, R = 64 * ceil(blocks / T) <= 17 856 nodes = 140 KB of u64), and a level is
//
//   consume   the 8-byte messages the T members wrote for me in the previous level: ds_add_u64 into my residue
//   sweep     my R residue words against their thresholds (algo.h:1012): whoever is at or over it is this level's
//             frontier -- after a level's pops every residue is under its threshold, so "crossed during the level" and
//             "is at or over it now" are the same set
//   pop+emit  batches of up to 1024 frontier nodes: residue -> reserve + increment (algo.h:983-1002, one lane per node),
//             then their out-edges from a copy of col that names every target as (owner, local id): one message
//             local | increment << 16 into the exact-capacity bucket (me -> owner), place taken from an LDS counter
//   barrier   ONE device-scope barrier of the team per level (arrive = atomic add of 2^32 + my pops on a word that
//             rotates over four; the sum of the pops is the level's frontier size, read by every member)
//
// No residue slab traffic, no frontier lists, no launches, no host round trips per level.  A bucket (s -> d) can never
// overflow: a level relaxes every edge at most once, so its capacity is the number of edges from s's nodes to d's
// (+ the few two-word messages, see below), counted when the graph is loaded.  Once a slot's frontier is small (and has
// been large), the members write their residue ranges to the slot's slab and the crossing nodes as (node, residue taken)
// entries to its frontier list: k_push_tail finishes all slots of the batch in one launch, as it does for the bucketed
// levels.  Same level-synchronous schedule, integer adds: bit-identical to the bucketed path and to oracle/fora_twin.c.
//
// Message word: local target (15 bits) | big (bit 15) | value << 16.  An increment of 2^48 or more (the first two or
// three levels; at most 2^14 of them fit the unit mass) travels as two words: (inc >> 14) with `big` set and
// (inc & 0x3fff) without.
#pragma once
#include "fora_kernels.h"

namespace fora {

constexpr int TEAM_MAX = 32;                  // members of a team (5 bits of a target word)
constexpr int TEAM_LBITS = 15;                // bits of a local id
constexpr int TEAM_THREADS = 1024;            // one workgroup per CU
constexpr int TEAM_NW = TEAM_THREADS / 64;
constexpr int TEAM_BATCH = 1024;              // frontier nodes popped per batch (one per thread)
constexpr int TEAM_NIT = 18;                  // sweep iterations at most: R <= 18 * 1024
constexpr uint32_t TEAM_R_CAP = 17792;        // local ids per member at most: 8 * R + the static LDS below <= 160 KiB
#ifndef FORA_TEAM_EPT
#define FORA_TEAM_EPT 8
#endif
constexpr int TEAM_EPT = FORA_TEAM_EPT;       // consecutive edges a lane gathers per chunk
constexpr uint32_t TEAM_BIG = 1u << 15;
constexpr uint32_t TEAM_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t ERR_TEAM_TIMEOUT = 16, ERR_TEAM_CAP = 32;

struct TeamDev {
    // the graph / slot state the kernel touches (a copy of the few Dev fields it needs: the 90-field struct by value costs
    // the kernel a hundred SGPRs it has to spill)
    int32_t n, nq;
    const uint64_t *rowinfo;
    const int64_t *row_ptr;
    const uint32_t *deg;
    const int32_t *src;
    uint64_t *residue, *ppr;
    uint32_t *fl0, *fl_count0;     // Dev::fl[0], Dev::fl_count[0]: the hand-over to k_push_tail
    uint64_t *inc_tab0;
    uint64_t segq_cap;
    QState *qs;
    uint32_t *err;
    uint64_t afix, t1;
    uint32_t T, R, nteams, nblk;   // members per team; local ids per member; teams of the launch; 64-node blocks of the graph
    const uint32_t *colt;          // [nnz] owner << 15 | local id of every edge target, rows as in col
    const uint32_t *off;           // [T * T + 1] first message slot of bucket (s -> d) at [s * T + d]; [T * T]: slots per (team, parity)
    uint64_t *msg;                 // [nteams][2][off[T * T]]
    uint32_t *cnt;                 // [nteams][2][T * T] messages in bucket (s -> d) this level
    unsigned long long *sync;      // [nteams][4][16] barrier words, one 128-byte line each
    uint32_t *slot_seq;            // [nteams][nq + 2] slot taken by the team in its k-th turn (TEAM_EMPTY: not yet)
    uint32_t *ctl;                 // [0] next slot, [32] abort flag
    uint32_t tail_max;             // hand the slot to k_push_tail once its frontier is at most this (and has been larger); 0: never
    uint32_t tail_always;          // tests: do not wait for the frontier to have been larger
    uint32_t xcd;                  // != 0: the members of a team share blockIdx % 8 (one XCD under round-robin placement: speed only)
    uint64_t timeout_ticks;        // wall_clock64 ticks (100 MHz) a member waits for its team before it gives up
};

__device__ __forceinline__ uint32_t team_deg(const TeamDev &a, uint64_t ri, uint32_t v) { // exact out-degree (see ri_deg)
    const uint32_t dg = (uint32_t)ri & DEG_SAT;
    return dg == DEG_SAT ? (uint32_t)(a.row_ptr[v + 1] - (int64_t)(ri >> 24)) : dg;
}
__device__ __forceinline__ uint32_t team_node(uint32_t l, uint32_t me, uint32_t T) { // local id -> node
    return ((((l >> 6) * T + me) << 6) | (l & 63u));
}

// one poll loop for everything a member waits for: returns false when the launch is being abandoned
template <class DONE>
__device__ __forceinline__ bool team_wait(const TeamDev &a, uint32_t *err, DONE done) {
    const uint64_t t0 = wall_clock64();
    for (uint32_t it = 1;; it++) {
        if (done()) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((it & 255u) == 0) {
            if (__hip_atomic_load(&a.ctl[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (wall_clock64() - t0 > a.timeout_ticks) {
                __hip_atomic_store(&a.ctl[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(err, ERR_TEAM_TIMEOUT);
                return false;
            }
        }
    }
}

// grid = nteams * T workgroups of TEAM_THREADS, all resident (one per CU); dynamic LDS = 8 * R bytes.
__global__ void __launch_bounds__(TEAM_THREADS) k_push_team(const TeamDev a) {
    extern __shared__ uint64_t res[];                   // [R] residue of my nodes
    __shared__ uint64_t s_inc[TEAM_BATCH];              // increment of batch entry i
    __shared__ uint32_t s_ebeg[TEAM_BATCH];             // its first edge
    __shared__ uint32_t s_pref[TEAM_BATCH + 1];         // exclusive prefix of the batch's out-degrees
    __shared__ uint16_t s_list[TEAM_BATCH];             // local ids of the batch
    __shared__ uint32_t s_cell[TEAM_NIT * TEAM_NW + 1]; // crossing nodes per (sweep iteration, wave), then their exclusive prefix
    __shared__ uint32_t s_w[TEAM_NW];
    __shared__ uint32_t s_fill[TEAM_MAX], s_moff[TEAM_MAX], s_mcap[TEAM_MAX]; // messages I have put into bucket (me -> d) this level; its first slot; its size
    __shared__ uint32_t s_cpre[TEAM_MAX + 1], s_coff[TEAM_MAX]; // prefix of the messages waiting for me per source; bucket (s -> me)
    __shared__ uint32_t s_slot, s_F, s_ok, s_base;
    __shared__ unsigned long long s_dang, s_acc[3];

    const uint32_t T = a.T, R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t team, me;
    if (a.xcd) { // blocks b and b + 8 share an XCD (observed, not promised): a team = T blocks of one residue class
        const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = gridDim.x >> 3; // per: blocks per class, a multiple of T
        team = x * (per / T) + j / T;
        me = j % T;
    } else {
        team = blockIdx.x / T;
        me = blockIdx.x % T;
    }
    const uint32_t nit = (R + TEAM_THREADS - 1) / TEAM_THREADS;
    const uint64_t cap_total = a.off[T * T];
    unsigned long long *sync = a.sync + (uint64_t)team * 4 * 16;
    uint32_t *seq = a.slot_seq + (uint64_t)team * ((uint32_t)a.nq + 2);
    for (uint32_t l = tid; l < R; l += TEAM_THREADS) res[l] = 0;
    if (tid < (int)T) { s_moff[tid] = a.off[me * T + tid]; s_mcap[tid] = a.off[me * T + tid + 1] - a.off[me * T + tid]; s_coff[tid] = a.off[tid * T + me]; }
    __syncthreads();
    uint32_t g = 0; // barriers this team has passed: message / count buffers by g & 1, barrier words by g & 3

    for (uint32_t turn = 0;; turn++) {
        // ---- the team's next slot: member 0 draws it, the others read it from the team's sequence
        if (tid == 0) {
            uint32_t s = TEAM_EMPTY;
            bool ok = true;
            if (me == 0) {
                s = atomicAdd(&a.ctl[0], 1u);
                __hip_atomic_store(&seq[turn], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                ok = team_wait(a, a.err, [&] { s = __hip_atomic_load(&seq[turn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return s != TEAM_EMPTY; });
            }
            s_slot = s; s_ok = ok ? 1u : 0u;
        }
        __syncthreads();
        if (!s_ok) return;
        const uint32_t q = s_slot;
        if (q >= (uint32_t)a.nq) break;
        const uint32_t src = (uint32_t)a.src[q];
        if (a.deg[src] == 0) continue; // dangling source: k_init_batch has written the whole answer (algo.h:961-965)
        const uint64_t slab = (uint64_t)q * a.n;
        const uint32_t src_owner = (src >> 6) % T, src_local = (((src >> 6) / T) << 6) | (src & 63u);
        uint64_t acc_res = 0, acc_pops = 0, acc_relax = 0;
        uint32_t peak = 0, nlev = 0;
        bool final_round = false;

        for (uint32_t L = 0;; L++) {
            // ================= consume: the messages of the previous level that are addressed to me
            if (L > 0) {
                const uint32_t *cin = a.cnt + ((uint64_t)team * 2 + ((g - 1) & 1u)) * T * T;
                const uint64_t *min_ = a.msg + ((uint64_t)team * 2 + ((g - 1) & 1u)) * cap_total;
                if (tid < 64) {
                    uint32_t c = (uint32_t)lane < T ? cin[(uint32_t)lane * T + me] : 0u;
                    uint32_t tot;
                    const uint32_t ex = wave_excl_scan(c, tot);
                    if ((uint32_t)lane < T) s_cpre[lane] = ex;
                    if (lane == 0) s_cpre[T] = tot;
                }
                __syncthreads();
                const uint32_t total = s_cpre[T];
                constexpr int CU = 4; // loads in flight per lane
                for (uint32_t i0 = 0; i0 < total; i0 += TEAM_THREADS * CU) {
                    uint64_t m[CU];
#pragma unroll
                    for (int k = 0; k < CU; k++) {
                        const uint32_t i = i0 + k * TEAM_THREADS + tid;
                        const uint32_t ic = i < total ? i : 0u;
                        uint32_t lo = 0, hi = T; // largest lo with s_cpre[lo] <= ic
#pragma unroll
                        for (int it = 0; it < 5; it++) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (hi - lo > 1) { if (s_cpre[mid] <= ic) lo = mid; else hi = mid; }
                        }
                        m[k] = total ? min_[(uint64_t)s_coff[lo] + (ic - s_cpre[lo])] : 0ull;
                    }
#pragma unroll
                    for (int k = 0; k < CU; k++) {
                        const uint32_t i = i0 + k * TEAM_THREADS + tid;
                        if (i < total) {
                            uint64_t val = m[k] >> 16;
                            if ((uint32_t)m[k] & TEAM_BIG) val <<= 14;
                            if (val) atomicAdd((unsigned long long *)&res[(uint32_t)m[k] & (TEAM_BIG - 1u)], (unsigned long long)val);
                        }
                    }
                }
                __syncthreads();
            }
            // ================= sweep: who is at or over the threshold (algo.h:1012)
            uint32_t crossmask = 0, ncross = 0;
            if (L == 0) { // the source is popped whatever its threshold (algo.h:969-978)
                if (tid == 0) { s_cell[0] = 0; s_cell[TEAM_NIT * TEAM_NW] = me == src_owner ? 1u : 0u; }
                if (me == src_owner && tid == 0) { res[src_local] = FIX_ONE; }
                __syncthreads();
                ncross = s_cell[TEAM_NIT * TEAM_NW];
            } else {
                constexpr int SG = 6; // iterations whose loads are in flight together
                for (uint32_t g0 = 0; g0 < nit; g0 += SG) {
                    uint64_t r[SG];
                    uint32_t dg[SG];
#pragma unroll
                    for (int k = 0; k < SG; k++) {
                        const uint32_t l = (g0 + k) * TEAM_THREADS + tid;
                        r[k] = ((uint32_t)(g0 + k) < nit && l < R) ? res[l] : 0ull;
                    }
#pragma unroll
                    for (int k = 0; k < SG; k++) {
                        dg[k] = 1;
                        if (r[k]) dg[k] = a.deg[team_node((g0 + k) * TEAM_THREADS + tid, me, T)];
                    }
#pragma unroll
                    for (int k = 0; k < SG; k++) {
                        const uint32_t it = g0 + k;
                        if (it < nit) { // wave-uniform
                            const bool c = r[k] && r[k] >= node_thr(a.t1, dg[k]);
                            const unsigned long long mk = __ballot(c);
                            if (lane == 0) s_cell[it * TEAM_NW + wid] = (uint32_t)__popcll(mk);
                            if (c) crossmask |= 1u << it;
                        }
                    }
                }
                __syncthreads();
                if (wid == 0) { // exclusive prefix over the (iteration, wave) cells
                    constexpr int CPL = (TEAM_NIT * TEAM_NW + 63) / 64;
                    const uint32_t cells = nit * TEAM_NW;
                    uint32_t cc[CPL], sum = 0;
#pragma unroll
                    for (int j = 0; j < CPL; j++) { cc[j] = (uint32_t)(CPL * lane + j) < cells ? s_cell[CPL * lane + j] : 0u; sum += cc[j]; }
                    uint32_t tot;
                    uint32_t ex = wave_excl_scan(sum, tot);
#pragma unroll
                    for (int j = 0; j < CPL; j++) { if ((uint32_t)(CPL * lane + j) < cells) s_cell[CPL * lane + j] = ex; ex += cc[j]; }
                    if (lane == 0) s_cell[TEAM_NIT * TEAM_NW] = tot;
                }
                __syncthreads();
                ncross = s_cell[TEAM_NIT * TEAM_NW];
            }
            if (final_round) {
                // ================= hand-over: crossing nodes -> (node, residue taken) entries of the slot's frontier list
                // (k_push_tail pops them, see k_accum), then my residue range -> the slot's slab
                if (ncross) {
                    if (tid == 0) s_base = atomicAdd(&a.fl_count0[q * CSTRIDE], ncross);
                    __syncthreads();
                    const uint32_t base = s_base;
                    for (uint32_t it = 0; it < nit; it++) {
                        {
                            const bool c = (crossmask >> it) & 1u;
                            const unsigned long long mk = __ballot(c);
                            if (c) {
                                const uint32_t l = it * TEAM_THREADS + tid;
                                const uint32_t pos = base + s_cell[it * TEAM_NW + wid] + (uint32_t)__popcll(mk & ((1ull << lane) - 1));
                                if (pos < (uint32_t)a.n) {
                                    a.fl0[slab + pos] = team_node(l, me, T);
                                    a.inc_tab0[(uint64_t)q * a.segq_cap + pos] = res[l];
                                } else atomicOr(a.err, ERR_WL_OVERFLOW);
                                res[l] = 0;
                            }
                        }
                    }
                }
                for (uint32_t l = tid; l < R; l += TEAM_THREADS) {
                    const uint32_t v = team_node(l, me, T);
                    if (v < (uint32_t)a.n) a.residue[slab + v] = res[l];
                    res[l] = 0;
                }
                break;
            }
            // ================= pop + emit, a batch of TEAM_BATCH frontier nodes at a time
            uint64_t *mout = a.msg + ((uint64_t)team * 2 + (g & 1u)) * cap_total;
            if (tid < TEAM_MAX) s_fill[tid] = 0;
            if (tid == 0) s_dang = 0;
            uint64_t my_dang = 0;
            for (uint32_t b0 = 0; b0 < ncross; b0 += TEAM_BATCH) {
                const uint32_t nb = min((uint32_t)TEAM_BATCH, ncross - b0);
                __syncthreads(); // the previous batch is done with the lists (and s_fill / s_dang are zeroed)
                if (L == 0) {
                    if (tid == 0) s_list[0] = (uint16_t)src_local;
                } else {
                    for (uint32_t it = 0; it < nit; it++) {
                        {
                            const bool c = (crossmask >> it) & 1u;
                            const unsigned long long mk = __ballot(c);
                            if (c) {
                                const uint32_t rank = s_cell[it * TEAM_NW + wid] + (uint32_t)__popcll(mk & ((1ull << lane) - 1)) - b0;
                                if (rank < (uint32_t)TEAM_BATCH) s_list[rank] = (uint16_t)(it * TEAM_THREADS + tid);
                            }
                        }
                    }
                }
                __syncthreads();
                uint32_t cnt = 0;
                if ((uint32_t)tid < nb) { // one pop per lane (algo.h:983-1002)
                    const uint32_t l = s_list[tid];
                    const uint32_t v = team_node(l, me, T);
                    const uint64_t rr = res[l];
                    res[l] = 0;                                       // algo.h:984-985
                    const uint64_t ri = a.rowinfo[v];
                    const uint32_t deg = team_deg(a, ri, v);
                    const uint64_t rsv_old = a.ppr[slab + v];
                    uint64_t rsv_add, dang;
                    const uint64_t inc = pop_value(a.afix, rr, deg, rsv_add, dang);
                    if (rsv_add) a.ppr[slab + v] = rsv_old + rsv_add; // algo.h:986-989 (this member owns v)
                    acc_res += rsv_add; my_dang += dang; acc_pops++; acc_relax += deg;
                    s_ebeg[tid] = (uint32_t)(ri >> 24);
                    s_inc[tid] = inc;
                    cnt = inc ? deg : 0u;
                }
                uint32_t total;
                const uint32_t pre = block_excl_scan_n<TEAM_THREADS>(cnt, s_w, total);
                s_pref[tid] = pre;
                if (tid == 0) s_pref[TEAM_BATCH] = total;
                __syncthreads();
                for (uint32_t cb = 0; cb < total; cb += TEAM_THREADS * TEAM_EPT) { // no barrier in here: the waves run free
                    const uint32_t e0 = cb + tid * TEAM_EPT;
                    if (e0 >= total) continue;
                    uint32_t lo = 0, hi = TEAM_BATCH;
#pragma unroll
                    for (int it = 0; it < 10; it++) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_pref[mid] <= e0) lo = mid; else hi = mid;
                    }
                    uint32_t w[TEAM_EPT], si[TEAM_EPT];
#pragma unroll
                    for (int k = 0; k < TEAM_EPT; k++) {
                        const uint32_t e = e0 + k;
                        if (e < total) while (s_pref[lo + 1] <= e) lo++; // entries without edges
                        si[k] = lo;
                    }
#pragma unroll
                    for (int k = 0; k < TEAM_EPT; k++) {
                        const uint32_t e = e0 + k;
                        w[k] = TEAM_EMPTY;
                        if (e < total) w[k] = a.colt[(uint64_t)s_ebeg[si[k]] + (e - s_pref[si[k]])];
                    }
#pragma unroll
                    for (int k = 0; k < TEAM_EPT; k++) {
                        if (w[k] == TEAM_EMPTY) continue;
                        const uint32_t dst = w[k] >> TEAM_LBITS, local = w[k] & (TEAM_BIG - 1u);
                        const uint64_t inc = s_inc[si[k]];
                        const bool big = (inc >> 48) != 0;
                        const uint32_t pos = atomicAdd(&s_fill[dst], big ? 2u : 1u);
                        uint64_t *at = mout + (uint64_t)s_moff[dst] + pos;
                        if (pos + (big ? 2u : 1u) > s_mcap[dst]) { atomicOr(a.err, ERR_TEAM_CAP); continue; } // cannot happen: the capacity is the bucket's edge count
                        if (!big) at[0] = (uint64_t)local | (inc << 16);
                        else { at[0] = (uint64_t)local | TEAM_BIG | ((inc >> 14) << 16); at[1] = (uint64_t)local | ((inc & 0x3FFFull) << 16); }
                    }
                }
            }
            // the dangling nodes' mass returns to the source within the level (algo.h:993-998): one message to its owner
            my_dang = wave_sum(my_dang);
            if (lane == 0 && my_dang) atomicAdd(&s_dang, (unsigned long long)my_dang);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave's message stores have completed before the barrier
            __syncthreads();
            if (tid == 0 && s_dang) {
                const uint64_t dm = s_dang;
                const bool big = (dm >> 48) != 0;
                const uint32_t pos = s_fill[src_owner];
                uint64_t *at = mout + (uint64_t)s_moff[src_owner] + pos;
                if (pos + 2u > s_mcap[src_owner]) atomicOr(a.err, ERR_TEAM_CAP);
                else if (!big) at[0] = (uint64_t)src_local | (dm << 16);
                else { at[0] = (uint64_t)src_local | TEAM_BIG | ((dm >> 14) << 16); at[1] = (uint64_t)src_local | ((dm & 0x3FFFull) << 16); }
                s_fill[src_owner] = pos + (big ? 2u : 1u);
            }
            __syncthreads();
            if ((uint32_t)tid < T) (a.cnt + ((uint64_t)team * 2 + (g & 1u)) * T * T)[me * T + tid] = s_fill[tid];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // ================= the team's barrier; the level's frontier size comes with it
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned long long *wd = &sync[(g & 3u) * 16];
                __hip_atomic_fetch_add(wd, (1ull << 32) | (unsigned long long)ncross, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long v = 0;
                const bool ok = team_wait(a, a.err, [&] { v = __hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return (uint32_t)(v >> 32) == T; });
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (me == 0) __hip_atomic_store(&sync[((g + 2) & 3u) * 16], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // dead since barrier g - 1
                s_F = (uint32_t)v;
                s_ok = ok ? 1u : 0u;
            }
            __syncthreads();
            if (!s_ok) return;
            g++;
            const uint32_t F = s_F; // nodes the team popped in this level
            if (F) nlev++;
            peak = max(peak, F);
            final_round = F == 0 || (a.tail_max && F <= a.tail_max && (peak > a.tail_max || a.tail_always));
        }
        // ---- the slot's counters (algo.h:992 rsum bookkeeping)
        acc_res = wave_sum(acc_res); acc_pops = wave_sum(acc_pops); acc_relax = wave_sum(acc_relax);
        if (tid < 3) s_acc[tid] = 0;
        __syncthreads();
        if (lane == 0 && acc_pops) {
            atomicAdd(&s_acc[0], (unsigned long long)acc_res);
            atomicAdd(&s_acc[1], (unsigned long long)acc_pops);
            atomicAdd(&s_acc[2], (unsigned long long)acc_relax);
        }
        __syncthreads();
        if (tid == 0) {
            QState *s = &a.qs[q];
            if (s_acc[1]) {
                atomicAdd(&s->reserved, s_acc[0]);
                atomicAdd(&s->pops, s_acc[1]);
                if (s_acc[2]) atomicAdd(&s->relax, s_acc[2]);
            }
            if (me == 0 && nlev) atomicAdd(&s->levels, nlev);
        }
    }
}

} // namespace fora
