// fora_team.h -- the "team push": forward push (algo.h:954-1018) with the residue of a slot resident in LDS for the
// whole push.
//
// The bucketed push of fora_kernels.h (k_pushq_bin + k_accum) keeps one level of increments in LDS and sends the
// residue HBM -> LDS sum -> HBM every level: at the mid-size levels of a ws-sized query that rewrite of the slot's slab
// is most of the bytes the pair moves.  Here a TEAM of T workgroups, one per CU, owns a slot from its
// first level to (nearly) its last.  Member c keeps the residue of the nodes it owns in LDS: 64-node blocks are dealt
// round-robin (owner(v) = (v >> 6) % T) and only nodes that HAVE in-edges get a local id (nothing ever lands on the
// others; the slot's source, if it is one of them, uses the spare id R).  A level is
//
//   consume   the 4-byte messages the members wrote for me in the previous level: ds_add_u64 into my residue.  There is
//             no barrier between the levels: a member ends its level by writing one tagged word per destination (count |
//             pops | level), and every wave of the destination polls its T words and consumes the buckets of the sources
//             that are through while the slower ones are still emitting
//   sweep     my residue words against their thresholds (algo.h:1012): whoever is at or over it is this level's frontier
//             -- after a level's pops every residue is under its threshold, so "crossed during the level" and "is at or
//             over it now" are the same set.  The out-degrees a thread compares with sit in its registers for the whole
//             launch (16 bits each): the sweep issues no global load
//   pop+emit  every wave on its own, 64 frontier nodes at a time: residue -> reserve log + increment (algo.h:983-1002,
//             one lane per node), then their out-edges from a copy of col that names every target as (owner, local id):
//             one returning LDS add gives a message its slot in the bucket (me -> owner), one store writes it
//
// No residue slab traffic, no frontier lists, no launches, no host round trips per level.  A bucket (s -> d) can never
// overflow: a level relaxes every edge at most once, so its capacity is the number of edges from s's nodes to d's
// (+ one for the level's dangling mass), counted when the graph is loaded.  Once a slot's frontier is small (and has
// been large), the members write their residue ranges to the slot's slab and the crossing nodes as (node, residue taken)
// entries to its frontier list: k_push_tail finishes all slots of the batch in one launch, as it does for the bucketed
// levels.  Same level-synchronous schedule, integer adds: bit-identical to the bucketed path and to oracle/fora_twin.c.
//
// Message word (4 bytes): local target (15 bits) | entry << 15, where `entry` names the popped node in the producer's
// increment table of the level (one 8-byte word per pop, written once, coalesced; the consumer gathers it from L2 --
// a bucket's messages follow the producer's pop order, so neighbouring lanes gather neighbouring entries).
//
// What was measured on the way: DESIGN.md 5.1b (ws-sized graph, 1000 queries: bucketed kernels 75 ms; first form of this
// kernel 89 ms; 49 ms now).
#pragma once
#include "fora_kernels.h"

namespace fora {

constexpr int TEAM_MAX = 32;                  // members of a team (5 bits of a target word)
constexpr int TEAM_LBITS = 15;                // bits of a local id
#ifndef FORA_TEAM_THREADS
#define FORA_TEAM_THREADS 1024
#endif
constexpr int TEAM_THREADS = FORA_TEAM_THREADS; // 1024: one workgroup per CU; 512: two per CU (members of two different teams: one team's waits overlap the other's work)
#ifndef FORA_TEAM_WGS_PER_CU
#define FORA_TEAM_WGS_PER_CU (FORA_TEAM_THREADS == 1024 ? 1 : 2)
#endif
constexpr int TEAM_WGS_PER_CU = FORA_TEAM_WGS_PER_CU;
constexpr int TEAM_NW = TEAM_THREADS / 64;
constexpr uint32_t TEAM_R_CAP = TEAM_WGS_PER_CU == 1 ? 15296 : 7680; // local ids per member at most: 8 * (R + 1) + the static LDS below <= 160 KiB / workgroups per CU
constexpr int TEAM_NIT = (TEAM_R_CAP + TEAM_THREADS - 1) / TEAM_THREADS; // sweep iterations at most (ids per thread): 15 (30 with 512 threads and one workgroup per CU)
#ifndef FORA_TEAM_EPT
#define FORA_TEAM_EPT 4
#endif
#ifndef FORA_TEAM_CU
#define FORA_TEAM_CU 4
#endif
#ifndef FORA_TEAM_MPL
#define FORA_TEAM_MPL 2 // messages a lane consumes per load: 2 (one 8-byte load); 4 (one 16-byte load through inline asm, 256-message segments) measured 46.7 against 45.8 ms
#endif
#ifndef FORA_TEAM_DRAW
#define FORA_TEAM_DRAW 4 // 64-id groups a wave draws at a time
#endif
constexpr int TEAM_EPT = FORA_TEAM_EPT;       // consecutive edges a lane gathers per chunk
constexpr int TEAM_CHUNK = 64 * TEAM_EPT;     // edges of a chunk
constexpr uint32_t TEAM_LMASK = (1u << TEAM_LBITS) - 1u;
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
    uint32_t T, R, nteams;         // members per team; local ids per member (a multiple of 64; id R: the slot's source when it has no in-edge); teams of the launch
    const uint32_t *colt;          // owner << 15 | local id of every edge target, rows as in col but every row padded with TEAM_EMPTY words to whole QUADS (4 words,
                                   // 16-byte aligned): a lane reads the four edges of a quad with one 16-byte load, and all four belong to one row
    const uint32_t *rowq;          // [n] first quad of a node's row (the slot's source when it has no local id)
    const uint32_t *n2l;           // [n] owner << 15 | local id of a node, TEAM_EMPTY for a node without in-edges
    const uint32_t *l2n;           // [T][R] node of a local id (TEAM_EMPTY: unused id)
    const uint16_t *deg16;         // [T][R] its out-degree, saturating at 0xFFFF (then Dev::deg has it)
    const uint64_t *rowl;          // [T][R] node (19 bits) | out-degree (13 bits, 8191: look it up) << 19 | first quad of its row in colt << 32: one load per pop, by LOCAL id
    uint64_t *rsvl;                // [nteams][T][R] reserve accumulators by local id (all zero between slots) for the pops that do not fit the member's log below
    // Reserve LOG: a pop's reserve (algo.h:986-989) is not added to the node's accumulator when it happens (a load and a
    // store of a random 8-byte word per pop: a fifth of the kernel's L2 requests) -- the member appends (local id, amount) to
    // its log of the slot with two coalesced stores, and at the hand-over, when its LDS no longer holds the residue, replays
    // the log into that LDS and writes the sums to the slot's ppr slab.  A log that is full sends the pop to rsvl as before.
    uint16_t *rlog_id;             // [nteams][T][rlog_cap]
    uint64_t *rlog_val;            // [nteams][T][rlog_cap]
    uint32_t rlog_cap;
    // Hub pre-aggregation: an edge whose target is one of the H nodes of largest in-degree (colt word 0x80000000 | hub) adds
    // its increment to the member's LDS sum of that hub; after the level's rows every non-zero sum leaves as ONE message
    // (its own entry of the increment table).  On the ws-sized graph 27 % of all edges end at the top 1024 nodes.
    uint32_t H;                    // hubs (0: none)
    const uint32_t *hubtgt;        // [H] owner << 15 | local id of hub h
    const uint32_t *off;           // [T * T + 1] first message slot of bucket (s -> d) at [s * T + d]; [T * T]: slots per (team, parity)
    uint32_t *msg;                 // [nteams][2][off[T * T]]
    uint64_t *inct;                // [nteams][2][T][R + 64 + H] increment tables: entry e of member s = the increment of its e-th pop of the level; behind the pops: the dangling mass, then the hub sums
    unsigned long long *cntw;      // [nteams][2][T * T] the members' words of a level, [destination][source]: messages in bucket (s -> d) (24 bits) | s's pops << 24 | level tag << 40; zero at launch
    unsigned long long *sync;      // [nteams][5][16] the fifth 128-byte line: the members' XCD census (the others: unused)
    uint32_t *slot_seq;            // [nteams][nq + 2] slot taken by the team in its k-th turn (TEAM_EMPTY: not yet)
    uint32_t *ctl;                 // [0] next slot, [32] abort flag
    uint32_t tail_max;             // hand the slot to k_push_tail once its frontier is at most this (and has been larger); 0: never
    uint32_t tail_always;          // tests: do not wait for the frontier to have been larger
    uint32_t xcd;                  // != 0: the members of a team share blockIdx % 8 (one XCD under round-robin placement: speed only; the kernel checks where they really are); 2: the same, but always with the fences (tests)
    unsigned long long *stamps;    // diagnostic builds (-DFORA_STAMPS): cycles per phase of thread 0, summed over workgroups, [0..7]
    uint64_t timeout_ticks;        // wall_clock64 ticks (100 MHz) a member waits for its team before it gives up
    uint32_t abort_level;          // tests (option team_abort_level): != 0: every member abandons the launch when its slot reaches this level, as after a time-out
};

// The kernel never keeps its argument struct in registers.  By value, the ~45 fields were loaded at the kernel's entry
// and lived in SGPRs -- 272 of them spilled into VGPR lanes, 1035 v_readlane / v_writelane of 5975 static instructions,
// five VGPR spills on top because those lanes occupied a kernel already at the 128-register cap (round 4's code object,
// tools/isa_audit.py).  Instead every PHASE reads what it needs from the kernarg segment (constant address space: scalar
// loads, cached) through a pointer it has just laundered: the loads cannot move above the laundering, so a field is live
// from its phase's start to its last use there and no longer.
typedef const __attribute__((address_space(4))) TeamDev *TeamArgs;
__device__ __forceinline__ TeamArgs team_args() {
    TeamArgs p = (TeamArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// set bits of a wave mask below this lane (v_mbcnt: two instructions, the mask in scalar registers)
__device__ __forceinline__ uint32_t rank_below(unsigned long long mk) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
}
// A per-thread value the optimiser must treat as new: predicates derived from it (tid < 32, lane == 0 && ...) are not
// hoisted out of the level loop, where each lived in two scalar registers for the whole launch (130 of them spilled).
__device__ __forceinline__ int fresh(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ uint32_t fresh_s(uint32_t x) { asm volatile("" : "+s"(x)); return x; } // (the same for a wave-uniform value)

// one poll loop for everything a member waits for: returns false when the launch is being abandoned
template <class DONE>
__device__ __forceinline__ bool team_wait(DONE done) {
    const uint64_t t0 = wall_clock64();
    {
        const TeamArgs a = team_args();
        if (a->timeout_ticks == 0) { // option team_timeout_ms = 0 (tests): give up at once, as if the team had waited in vain
            __hip_atomic_store(&a->ctl[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(a->err, ERR_TEAM_TIMEOUT);
            return false;
        }
    }
    for (uint32_t it = 1;; it++) {
        if (done()) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((it & 255u) == 0) {
            const TeamArgs a = team_args();
            uint32_t *ctl = a->ctl;
            if (__hip_atomic_load(&ctl[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (wall_clock64() - t0 > a->timeout_ticks) {
                __hip_atomic_store(&ctl[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(a->err, ERR_TEAM_TIMEOUT);
                return false;
            }
        }
    }
}

// (TSTAMP* phase stamps and the marginal-cost probes: fora_diag.h -- empty in the product build)

#ifndef FORA_TEAM_HEAVY
#define FORA_TEAM_HEAVY 1024
#endif
constexpr uint32_t TEAM_HEAVY = FORA_TEAM_HEAVY; // rows of more edges are relaxed by the whole workgroup, consecutive lanes on consecutive edges
#ifndef FORA_TEAM_NHEAVY
#define FORA_TEAM_NHEAVY 128
#endif
#ifndef FORA_TEAM_MAXGROUPS
#define FORA_TEAM_MAXGROUPS 256
#endif
constexpr int TEAM_NHEAVY = FORA_TEAM_NHEAVY;       // heavy rows of a level a member can share out (the others stay with the wave that popped them)
constexpr int TEAM_MAXGROUPS = FORA_TEAM_MAXGROUPS; // 64-id groups of a member at most, the spare id's included

// word[k] / dst[k]: this lane's messages of a chunk (dst TEAM_EMPTY: none).  s_fill[d]: the next free slot of my bucket
// (me -> d) in the level's message buffer -- one returning LDS add per message; all four in flight together, then the stores.
// (Measured twice and dropped, rounds 4 and 5: the chunk's messages sorted by destination in a wave-private LDS stage --
// counting sort over the <= 32 destinations, one fill-counter atomic per (chunk, destination) -- and stored with consecutive
// lanes on consecutive words of a run: 82-84 ms against 75 in round 4, 51.9 against 50.9 in round 5 (both without hub sums,
// whose LDS the stage needs): the stage's LDS round trips cost what the 4 x fewer write requests save.
// Round 6, a third form without any sort: per destination a RING of 64-word blocks in LDS (16 KB), a message written to the ring
// entry of its bucket position, the block's 64th writer (a returning LDS count) flushing it with one coalesced 256-byte store, ring
// blocks re-armed by their flush -- bit-exact (24 team tests), 82.9 ms against 45.8: three more LDS operations per message and
// waves waiting for ring blocks cost twice what the scattered stores do.  Messages leave the CU as scattered 4-byte stores.)
__device__ __forceinline__ void team_emit(const uint32_t (&word)[TEAM_EPT], const uint32_t (&dst)[TEAM_EPT], uint32_t *s_fill, uint32_t *mout) {
    uint32_t slot[TEAM_EPT];
#pragma unroll
    for (int k = 0; k < TEAM_EPT; k++) {
        slot[k] = 0;
        if (dst[k] != TEAM_EMPTY) slot[k] = atomicAdd(&s_fill[dst[k]], 1u);
    }
#pragma unroll
    for (int k = 0; k < TEAM_EPT; k++)
        if (dst[k] != TEAM_EMPTY) mout[slot[k]] = word[k];
    diag::team_emit_probe(word, dst, slot, mout, TEAM_EMPTY); // (nothing in the product build)
}

// grid = nteams * T workgroups of TEAM_THREADS, all resident (one per CU); dynamic LDS = 8 * (R + 1) bytes.
//
// Inside a level nothing but the consume -> sweep and the emit -> barrier seams is a workgroup barrier: the waves draw the
// level's 64-id groups from a shared counter, collect the crossing nodes in a wave-private list and pop / emit them 64 at a
// time on their own (prefix sums by wave scan), so the sixteen waves of a member overlap each other's memory round trips.
__global__ void __launch_bounds__(TEAM_THREADS, TEAM_THREADS * TEAM_WGS_PER_CU / 256) k_push_team(const TeamDev kernarg_only) { // (4 waves per SIMD: one 1024-thread or two 512-thread workgroups per CU; the argument is read through team_args())
    extern __shared__ uint64_t res[];                        // [R + 1] residue of my nodes; [R]: the slot's source when it has no local id; then [H] hub sums of the level
    __shared__ uint16_t w_list[TEAM_NW][128];                // per wave: local ids of crossing nodes waiting to be popped
    __shared__ unsigned long long s_gmask[TEAM_MAXGROUPS];   // crossing nodes of every 64-id group of the level
    __shared__ uint32_t h_ent[TEAM_NHEAVY], h_ebeg[TEAM_NHEAVY], h_deg[TEAM_NHEAVY], h_cstart[TEAM_NHEAVY]; // heavy rows of the level: table entry, first edge, degree (written last: 0 = not there yet), first chunk number
    __shared__ uint8_t w_mark[TEAM_NW][64] __attribute__((aligned(4))); // per wave: row marks of a chunk's quads
    __shared__ uint64_t w_inc[TEAM_NW][64];                  // per wave: increments of the nodes of its batch (hub edges add them in LDS)
    __shared__ uint64_t h_inc[TEAM_NHEAVY];
    __shared__ uint32_t s_hubent;
    __shared__ uint32_t s_hchunks, s_hnext, s_wdone; // chunks the heavy rows have been cut into so far; next one to take; waves done with their own rows
    __shared__ uint32_t s_fill[TEAM_MAX], s_moff[TEAM_MAX];  // next free slot of my bucket (me -> d) in the level's message buffer; its first slot
    __shared__ uint32_t s_slot, s_F, s_ok, s_ncross, s_nheavy, s_gnext, s_abort, s_rsvovf;
    __shared__ unsigned long long s_dang, s_acc[3];
    (void)kernarg_only;

    // What stays in scalar registers for the whole launch: T, R, H, my place (team, me), the size of a message buffer.
    uint32_t T, R, H, team, me, cap_total, coffv;
    const int tid0 = threadIdx.x;
    const int tid = tid0, lane = tid & 63, wid = tid >> 6;
    // out-degrees of the local ids this thread sweeps (it * 1024 + tid), 16 bits each: they never change
    uint32_t dgp[(TEAM_NIT + 1) / 2], thh[TEAM_NIT];
    bool same_xcd;
    uint32_t hub_tg0; // owner << 15 | local id of hub `tid` (the first TEAM_THREADS hubs: a register instead of a load at every level's end)
    {
        const TeamArgs a = team_args();
        T = a->T; R = a->R; H = a->H;
        hub_tg0 = (uint32_t)tid0 < H ? a->hubtgt[tid0] : 0u;
        if (a->xcd) { // blocks b and b + 8 share an XCD (observed, not promised): a team = T blocks of one residue class
            const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = gridDim.x >> 3; // per: blocks per class, a multiple of T
            team = x * (per / T) + j / T;
            me = j % T;
        } else {
            team = blockIdx.x / T;
            me = blockIdx.x % T;
        }
        const uint32_t *off = a->off;
        cap_total = off[T * T];
        for (uint32_t l = tid; l <= R + H; l += TEAM_THREADS) res[l] = 0; // (residues and hub sums)
        if (tid < TEAM_MAX) s_moff[tid] = tid < (int)T ? off[me * T + tid] : 0u;
        coffv = (uint32_t)lane < T ? off[(uint32_t)lane * T + me] : 0u; // first slot of bucket (lane -> me)
        const uint16_t *deg16 = a->deg16 + (uint64_t)me * R;
        const uint32_t wbase = (uint32_t)__builtin_amdgcn_readfirstlane(wid) * 64u;
#pragma unroll
        for (int it = 0; it < TEAM_NIT; it++) { // (R is a multiple of 64: a wave's 64 ids are all below R or none is)
            const uint32_t l = it * TEAM_THREADS + tid;
            const bool in = it * TEAM_THREADS + wbase < R;
            const uint32_t dv = in ? (uint32_t)deg16[l] : 0u;
            if (it & 1) dgp[it >> 1] |= dv << 16; else dgp[it >> 1] = dv;
            // ... and the HIGH WORD of its threshold (algo.h:1012: t1 x out-degree, the exact degree for a hub): the sweep compares
            // a residue's high word with it and only looks closer when the two are equal
            uint32_t dx = dv;
            if (in && dv == 0xFFFFu) dx = a->deg[a->l2n[(uint64_t)me * R + l]];
            thh[it] = in ? (uint32_t)(node_thr(a->t1, dx) >> 32) : 0xFFFFFFFFu;
        }
        // Do the team's members share an XCD?  Each adds 1 to the byte of its XCC id (HW_REG_XCC_ID) in the census word and
        // waits for all T.  Members of one XCD share its L2: what a member has stored (and waited for: vmcnt(0)) is in that L2,
        // and a load that bypasses L1 (sc1: relaxed agent-scope atomic load) sees it -- no release write-back of the L2 and no
        // acquire invalidate per level (~1.7 us each and more with freshly dirtied lines, MI355X guide).  Otherwise: both fences.
        if (tid == 0) {
            s_abort = 0;
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long *cw = a->sync + (uint64_t)team * 5 * 16 + 4 * 16;
            __hip_atomic_fetch_add(cw, 1ull << (8 * (xcc & 7u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long v = 0;
            const bool ok = team_wait([&] {
                v = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t tot = 0;
                for (int i = 0; i < 8; i++) tot += (uint32_t)(v >> (8 * i)) & 0xFFu;
                return tot == T;
            });
            bool one = false;
            for (int i = 0; i < 8; i++) one |= ((uint32_t)(v >> (8 * i)) & 0xFFu) == T;
            s_F = (one && a->xcd != 2) ? 1u : 0u; // (xcd == 2: tests force the fenced form)
            s_ok = ok ? 1u : 0u;
        }
        __syncthreads();
        if (!s_ok) return;
        same_xcd = s_F != 0;
        __syncthreads();
    }
    const uint32_t nit = (R + TEAM_THREADS - 1) / TEAM_THREADS; // <= TEAM_NIT
    const uint32_t ngroups = R / 64 + 1;                        // the last one holds the spare id R alone
    const uint32_t tstride = R + 64 + H;                        // entries of one increment table
    unsigned long long *s_hub = (unsigned long long *)(res + R + 1); // [H]
    TSTAMP_DECL
    uint32_t g = 0; // barriers this team has passed: message / count buffers by g & 1, barrier words by g & 3

    for (uint32_t turn = 0;; turn++) {
        // ---- the team's next slot: member 0 draws it, the others read it from the team's sequence
        uint32_t q, src, src_deg, src_owner, src_local;
        bool src_spare; // the source has no in-edge: it uses the spare id R
        uint64_t slab;
        {
            const TeamArgs a = team_args();
            const uint32_t nq = (uint32_t)a->nq;
            if (fresh(tid0) == 0) {
                uint32_t *seq = a->slot_seq + (uint64_t)team * (nq + 2);
                uint32_t s = TEAM_EMPTY;
                bool ok = true;
                if (fresh_s(me) == 0) {
                    s = atomicAdd(&a->ctl[0], 1u);
                    __hip_atomic_store(&seq[turn], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    ok = team_wait([&] { s = __hip_atomic_load(&seq[turn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return s != TEAM_EMPTY; });
                }
                s_slot = s; s_ok = ok ? 1u : 0u;
            }
            __syncthreads();
            if (!s_ok) return;
            q = s_slot;
            TSTAMP(7);
            if (q >= nq) break;
            src = (uint32_t)a->src[q];
            src_deg = a->deg[src];
            if (src_deg == 0) continue; // dangling source: k_init_batch has written the whole answer (algo.h:961-965)
            slab = (uint64_t)q * (uint32_t)a->n;
            src_owner = (src >> 6) % T;
            const uint32_t src_word = a->n2l[src];
            src_spare = src_word == TEAM_EMPTY;
            src_local = src_spare ? R : (src_word & TEAM_LMASK); // no in-edge: the spare id
        }
        uint64_t acc_res = 0, acc_pops = 0, acc_relax = 0;
        uint32_t peak = 0, nlev = 0;
        uint32_t logbase = 0;     // my pops of the slot's levels so far = entries of my reserve log
        bool final_round = false;
        if (fresh(tid0) == 0) s_rsvovf = 0;

        TSTAMP_LEVEL_DECL
        for (uint32_t L = 0;; L++) {
            const int tid = fresh(tid0), lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6); // this level's copies (see fresh())
            // ================= consume: the messages of the previous level that are addressed to me
            if (L > 0) {
                const TeamArgs a = team_args();
                const uint32_t par = (g - 1) & 1u;
                const uint32_t *min_ = a->msg + ((uint64_t)team * 2 + par) * cap_total;
                const uint64_t *tin = a->inct + ((uint64_t)team * 2 + par) * T * tstride;
                // There is no barrier between the levels: member s ends a level by writing ONE word per destination d -- the
                // messages in bucket (s -> d) | its pops << 24 | a tag of the level's number << 40 -- and every wave of d polls
                // the T words addressed to d and consumes the buckets of the sources that are through, while the slower ones
                // are still emitting (first form: a barrier of the team, then the counts: the members waited 12 % of their
                // cycles for the slowest one with their own input sitting ready).
                // A wave takes the 128-message segments j of source s with (j + s) % 16 == its number: balanced whether the
                // buckets are long (peak levels) or hold one segment each, and which (source, segment) its i-th one is follows
                // from a scan of the per-source counts and a ballot -- no table, no search.  CU segments per trip: all message
                // loads in flight together (two messages per lane), then all increment gathers.
                const uint32_t tagp = g & 0xFFFFFFu; // (the tag of barrier g - 1)
                const unsigned long long *cwin = a->cntw + ((uint64_t)team * 2 + par) * T * T + (uint64_t)me * T;
                const unsigned long long full = T >= 64 ? ~0ull : (1ull << T) - 1ull;
                unsigned long long donemask = 0;
                uint32_t fsum = 0, spins = 0;
                uint64_t w_t0 = 0;
                while (donemask != full) {
                unsigned long long wv = 0;
                if ((uint32_t)lane < T) wv = __hip_atomic_load(&cwin[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (sc1: past L1, like every load of handed-over data)
                const unsigned long long ready = __ballot((uint32_t)lane < T && (uint32_t)(wv >> 40) == tagp) & ~donemask;
                if (!ready) {
                    TSTAMP_SLEEP(8, 2, 40); // s_sleep(2) (stamps builds: charged to slot 8 as pure waiting inside the consume, + the poll itself)
                    if ((++spins & 255u) == 0) { // (rare: its operands are read here, not held through the loop)
                        const TeamArgs ar = team_args();
                        uint32_t *ctl = ar->ctl;
                        const uint64_t now = wall_clock64();
                        if (w_t0 == 0) w_t0 = now;
                        bool stop = __hip_atomic_load(&ctl[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (!stop && now - w_t0 > ar->timeout_ticks) {
                            __hip_atomic_store(&ctl[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (lane == 0) atomicOr(ar->err, ERR_TEAM_TIMEOUT);
                            stop = true;
                        }
                        if (stop) { if (lane == 0) s_abort = 1u; break; }
                    }
                    continue;
                }
                // (Measured in round 5 and dropped: ONE L1 invalidate per look -- buffer_inv sc1 -- and plain loads through the L1 for
                // the messages and the increment gathers instead of loads past it: 46.9 -> 121.9 ms; the invalidate stalls the CU's
                // whole vector memory path every time.)
                if (!same_xcd) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                donemask |= ready;
                const bool mine = (ready >> lane) & 1ull;
                const uint32_t c = mine ? (uint32_t)wv & 0xFFFFFFu : 0u; // lane s: the messages source s has for me
                fsum += mine ? (uint32_t)(wv >> 24) & 0xFFFFu : 0u;
                // a lane takes MPL consecutive messages of a segment: 2 = one 8-byte load (default); 4 = one 16-byte load and half the
                // load instructions, measured slower (the wait for the asm loads serialises the trip; 256-message segments); CU segments per trip
                constexpr int MPL = FORA_TEAM_MPL, SEGSH = MPL == 4 ? 8 : 7;
                constexpr int CU = MPL == 4 ? (FORA_TEAM_CU + 1) / 2 : FORA_TEAM_CU;
                const uint32_t ns = (c + (1u << SEGSH) - 1u) >> SEGSH;
                const uint32_t r0 = ((uint32_t)wid + 2u * TEAM_NW - (uint32_t)lane % TEAM_NW) % TEAM_NW; // my first segment of source `lane`
                const uint32_t kmine = ns > r0 ? (ns - r0 + TEAM_NW - 1) / TEAM_NW : 0u;
                uint32_t nmine;
                const uint32_t pmine = wave_excl_scan(kmine, nmine);
                for (uint32_t i0 = 0; i0 < nmine; i0 += CU) {
                    uint32_t mw[CU][MPL];
                    uint4 mq[CU];
                    uint32_t srcm[CU], left[CU];
                    const uint32_t *mp[CU];
#pragma unroll
                    for (int k = 0; k < CU; k++) {
                        const bool have = i0 + k < nmine;              // (wave-uniform)
                        const uint32_t ic = have ? i0 + k : 0u;
                        const uint32_t sidx = (uint32_t)__popcll(__ballot(pmine <= ic)) - 1u; // the last source whose segments start at or before my ic-th (lanes >= T: pmine = nmine > ic)
                        const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)r0, (int)sidx) + (ic - (uint32_t)__builtin_amdgcn_readlane((int)pmine, (int)sidx)) * TEAM_NW;
                        const uint32_t n_s = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)sidx);
                        const uint32_t idx = (j << SEGSH) + (uint32_t)MPL * lane;
                        left[k] = (have && idx < n_s) ? n_s - idx : 0u; // messages of this lane's group that exist (0 .. MPL, or more = MPL)
                        srcm[k] = sidx;
                        mp[k] = min_ + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)coffv, (int)sidx) + (left[k] ? idx : 0u); // (buckets start on and hold a multiple of 16 words)
                        if (MPL != 4) {
                            const unsigned long long mm = __hip_atomic_load((const unsigned long long *)mp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            mw[k][0] = (uint32_t)mm; mw[k][1] = (uint32_t)(mm >> 32);
                        }
                    }
                    if (MPL == 4) {
                        // 16-byte loads past the L1 (sc1, like every load of handed-over data).  The compiler neither issues nor counts
                        // them: loads and the wait for them are ONE asm statement, nothing reads a destination register before it
                        static_assert(MPL != 4 || CU <= 2, "one asm statement per trip");
                        if (CU == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(mq[0]) : "v"(mp[0]) : "memory");
                        else asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                          : "=&v"(mq[0]), "=&v"(mq[CU - 1]) : "v"(mp[0]), "v"(mp[CU - 1]) : "memory");
#pragma unroll
                        for (int k = 0; k < CU; k++) { mw[k][0] = mq[k].x; mw[k][1] = mq[k].y; mw[k][MPL - 2] = mq[k].z; mw[k][MPL - 1] = mq[k].w; }
                    }
                    uint64_t vv[CU][MPL];
#pragma unroll
                    for (int k = 0; k < CU; k++) {
                        const uint64_t *tb = tin + (uint64_t)srcm[k] * tstride;
#pragma unroll
                        for (int u = 0; u < MPL; u++)
                            vv[k][u] = __hip_atomic_load((const unsigned long long *)&tb[left[k] > (uint32_t)u ? (mw[k][u] >> TEAM_LBITS) : 0u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int k = 0; k < CU; k++) // (fora_diag.h: one more gather per message in the gather-probe build, nothing otherwise)
#pragma unroll
                        for (int u = 0; u < MPL; u++)
                            vv[k][u] += diag::team_gather_probe(tin + (uint64_t)srcm[k] * tstride, mw[k][u] >> TEAM_LBITS, left[k] > (uint32_t)u);
#pragma unroll
                    for (int k = 0; k < CU; k++)
#pragma unroll
                        for (int u = 0; u < MPL; u++)
                            if (left[k] > (uint32_t)u && vv[k][u]) atomicAdd((unsigned long long *)&res[mw[k][u] & TEAM_LMASK], (unsigned long long)vv[k][u]);
                }
                } // (sources that were through at this look)
                if (wid == 0) {
                    const uint32_t fs = wave_incl_scan_add(fsum);
                    if (lane == 63) s_F = fs; // nodes the team popped in the level before
                }
                __syncthreads();
                TSTAMP_GATE(s_F);
                TSTAMP(0);
                const TeamArgs af = team_args();
                bool quit = s_abort != 0;
                if (af->abort_level && L == af->abort_level) { // tests (option team_abort_level): give up in the middle of a slot, as after a time-out
                    if (tid == 0) { __hip_atomic_store(&af->ctl[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicOr(af->err, ERR_TEAM_TIMEOUT); }
                    quit = true;
                }
                if (quit) return;
                const uint32_t F = s_F;
                if (F) nlev++;
                peak = max(peak, F);
                const uint32_t tail_max = af->tail_max;
                final_round = F == 0 || (tail_max && F <= tail_max && (peak > tail_max || af->tail_always));
            }
            // ================= sweep: who is at or over the threshold (algo.h:1012).  Thread t looks at local ids
            // it * 1024 + t, i.e. wave w at the 64-id groups it * 16 + w; no global load (out-degrees in registers).
            uint32_t crossmask = 0;
            if (tid < TEAM_MAX) s_fill[tid] = s_moff[tid];
            if (tid == 0) { s_dang = 0; s_ncross = 0; s_nheavy = 0; s_gnext = 0; s_hchunks = 0; s_hnext = 0; s_wdone = 0; s_hubent = 0; }
            if (tid < TEAM_NHEAVY) h_deg[tid] = 0;
            if (L > 0) {
                const TeamArgs a = team_args();
                const uint64_t t1 = a->t1;
                const bool thr_small = (t1 >> 47) == 0;                    // then a 16-bit degree's threshold is two multiplies
                const uint32_t t1_lo = (uint32_t)t1, t1_hi = (uint32_t)(t1 >> 32);
                constexpr int SG = 5; // LDS reads in flight together
#pragma unroll
                for (int g0 = 0; g0 < TEAM_NIT; g0 += SG) {
                    if ((uint32_t)g0 * TEAM_THREADS + (uint32_t)wid * 64u < R) { // (scalar: R is a multiple of 64, a wave's ids are all below it or none is)
                        uint64_t r[SG];
#pragma unroll
                        for (int k = 0; k < SG; k++) {
                            const uint32_t l = (g0 + k) * TEAM_THREADS + tid;
                            r[k] = ((uint32_t)(g0 + k) * TEAM_THREADS + (uint32_t)wid * 64u < R) ? res[l] : 0ull;
                        }
#pragma unroll
                        for (int k = 0; k < SG; k++) {
                            const int it = g0 + k;
                            if ((uint32_t)it * TEAM_THREADS + (uint32_t)wid * 64u < R) { // scalar
                                bool c = false;
                                const uint32_t rhi = (uint32_t)(r[k] >> 32);
                                if (rhi > thh[it]) c = true; // (most ids: one compare says no, a second says yes)
                                else if (rhi == thh[it] && r[k]) { // the high words agree (or the node is dangling: threshold 1): the whole product
                                    uint32_t dg = (dgp[it >> 1] >> ((it & 1) * 16)) & 0xFFFFu;
                                    asm volatile("" : "+v"(dg)); // (or the compiler keeps fifteen 64-bit thresholds per thread across the levels -- and spills them)
                                    if (dg == 0xFFFFu) { // a hub: its exact degree
                                        uint32_t li = it * TEAM_THREADS + tid;
                                        asm volatile("" : "+v"(li)); // (keeps the fifteen addresses of this rare path out of the level loop's registers)
                                        const TeamArgs ar = team_args();
                                        c = r[k] >= node_thr(t1, ar->deg[ar->l2n[(uint64_t)me * R + li]]);
                                    }
                                    else if (thr_small) c = r[k] >= (uint64_t)t1_lo * dg + ((uint64_t)(t1_hi * dg) << 32); // t1 < 2^47, dg < 2^16: no overflow (dg 0: any residue crosses)
                                    else c = r[k] >= node_thr(t1, dg);
                                }
                                if (c) crossmask |= 1u << it;
                                const unsigned long long mk = __ballot(c);
                                if (lane == 0) s_gmask[it * TEAM_NW + wid] = mk; // (group it * 16 + wid < R / 64: not the spare id's)
                            }
                        }
                    }
                }
                if (tid == 0) { // the spare id (the source, if it has no in-edge: only dangling mass ever lands there)
                    const uint64_t rs = res[R];
                    s_gmask[ngroups - 1] = (me == src_owner && rs && rs >= node_thr(t1, src_deg)) ? 1ull : 0ull;
                }
            } else if (me == src_owner && tid == 0) { // the source is popped whatever its threshold (algo.h:969-978)
                res[src_local] = FIX_ONE;
            }
            TSTAMP(1);
            __syncthreads(); // s_fill / s_dang / counters are zeroed, the masks are in place; (level 0) so is the source's residue
            if (final_round) {
                // ================= hand-over: crossing nodes -> (node, residue taken) entries of the slot's frontier list
                // (k_push_tail pops them, see k_accum), then my residue range -> the slot's slab
                const TeamArgs a = team_args();
                const uint32_t n = (uint32_t)a->n;
                const uint32_t *l2n = a->l2n + (uint64_t)me * R;
                uint32_t *fl0 = a->fl0 + slab;
                uint64_t *inc0 = a->inc_tab0 + (uint64_t)q * a->segq_cap;
                uint64_t *residue = a->residue + slab;
                uint32_t mine = 0;
                for (uint32_t it = 0; it < nit; it++) mine += (uint32_t)__popcll(__ballot((crossmask >> it) & 1u));
                const bool spare = wid == 0 && s_gmask[ngroups - 1] != 0; // (wave-uniform)
                if (spare) mine++;
                if (mine) { // wave-uniform
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&a->fl_count0[q * CSTRIDE], mine);
                    base = __shfl(base, 0);
                    if (spare) {
                        if (lane == 0) {
                            if (base < n) { fl0[base] = src; inc0[base] = res[R]; }
                            else atomicOr(a->err, ERR_WL_OVERFLOW);
                            res[R] = 0;
                        }
                        base++;
                    }
                    for (uint32_t it = 0; it < nit; it++) {
                        const bool c = (crossmask >> it) & 1u;
                        const unsigned long long mk = __ballot(c);
                        if (c) {
                            const uint32_t l = it * TEAM_THREADS + tid;
                            const uint32_t pos = base + (uint32_t)rank_below(mk);
                            if (pos < n) {
                                fl0[pos] = l2n[l];
                                inc0[pos] = res[l];
                            } else atomicOr(a->err, ERR_WL_OVERFLOW);
                            res[l] = 0;
                        }
                        base += (uint32_t)__popcll(mk);
                    }
                }
                { // (same thread as above for every l)  All node ids of the thread's local ids first, then the stores: one round trip for the
                  // whole range instead of one per 1024 ids (round 6: the hand-over was a chain of ~30 dependent trips per slot)
                    uint32_t lv[TEAM_NIT];
#pragma unroll
                    for (int it = 0; it < TEAM_NIT; it++) {
                        const bool in = (uint32_t)it * TEAM_THREADS + (uint32_t)wid * 64u < R; // (scalar)
                        lv[it] = in ? l2n[it * TEAM_THREADS + tid] : TEAM_EMPTY;
                    }
#pragma unroll
                    for (int it = 0; it < TEAM_NIT; it++) {
                        if ((uint32_t)it * TEAM_THREADS + (uint32_t)wid * 64u < R) {
                            const uint32_t l = it * TEAM_THREADS + tid;
                            if (lv[it] != TEAM_EMPTY) residue[lv[it]] = res[l];
                            res[l] = 0;
                        }
                    }
                }
                if (tid == 0) { // (the same thread that may have handed the spare id over)
                    if (me == src_owner && src_spare) residue[src] = res[R];
                    res[R] = 0;
                }
                __syncthreads();
                // the reserve log of the slot -> sums per local id in the (now empty) LDS -> the slot's ppr slab (zero there so far)
                const uint32_t rlog_cap = a->rlog_cap;
                const uint16_t *rlog_id = a->rlog_id + ((uint64_t)team * T + me) * rlog_cap;
                const uint64_t *rlog_val = a->rlog_val + ((uint64_t)team * T + me) * rlog_cap;
                const uint32_t nlog = min(logbase, rlog_cap);
                for (uint32_t i0 = 0; i0 < nlog; i0 += 8 * TEAM_THREADS) { // (eight entries per thread in flight)
                    uint64_t val[8];
                    uint32_t id[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const uint32_t i = i0 + k * TEAM_THREADS + tid;
                        val[k] = i < nlog ? rlog_val[i] : 0ull;
                        id[k] = i < nlog ? rlog_id[i] : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (val[k]) atomicAdd((unsigned long long *)&res[id[k]], (unsigned long long)val[k]);
                }
                __syncthreads();
                const bool ovf = s_rsvovf != 0;
                uint64_t *rsvl = a->rsvl + ((uint64_t)team * T + me) * R;
                uint64_t *ppr = a->ppr + slab;
                {
                    uint32_t lv[TEAM_NIT];
#pragma unroll
                    for (int it = 0; it < TEAM_NIT; it++) {
                        const bool in = (uint32_t)it * TEAM_THREADS + (uint32_t)wid * 64u < R; // (scalar)
                        lv[it] = in ? l2n[it * TEAM_THREADS + tid] : TEAM_EMPTY;
                    }
#pragma unroll
                    for (int it = 0; it < TEAM_NIT; it++) {
                        if ((uint32_t)it * TEAM_THREADS + (uint32_t)wid * 64u < R) {
                            const uint32_t l = it * TEAM_THREADS + tid;
                            uint64_t rs = res[l];
                            if (ovf) { const uint64_t o = rsvl[l]; if (o) { rs += o; rsvl[l] = 0; } }
                            if (rs) { ppr[lv[it]] = rs; res[l] = 0; }
                        }
                    }
                }
                TSTAMP(6);
                break;
            }
            // ================= pop + emit, every wave on its own.  The waves draw the level's 64-id groups four at a time
            // from a shared counter (a group whose nodes have long rows keeps one wave busy while the others take the
            // rest), collect the crossing nodes in a wave-private list and pop 64 of them at a time.
            uint64_t my_dang = 0;
            {
                const TeamArgs a = team_args();
                uint32_t *mout = a->msg + ((uint64_t)team * 2 + (g & 1u)) * cap_total;
                uint64_t *tout = a->inct + (((uint64_t)team * 2 + (g & 1u)) * T + me) * tstride; // my increment table of this level
                const uint64_t *rowl = a->rowl + (uint64_t)me * R;
                const uint32_t rlog_cap = a->rlog_cap;
                uint16_t *rlog_id = a->rlog_id + ((uint64_t)team * T + me) * rlog_cap;
                uint64_t *rlog_val = a->rlog_val + ((uint64_t)team * T + me) * rlog_cap;
                const uint64_t afix = a->afix;
                const uint32_t *colt = a->colt;
                uint16_t *list = w_list[wid];
                uint64_t *winc = w_inc[wid];
                uint8_t *mark = w_mark[wid];
                uint32_t npend = 0, gcur = 0, gend = 0; // groups gcur .. gend - 1 of my current draw are not looked at yet
                bool drained = false;
                // (Measured in round 5 and dropped: batches of frontier / 16 nodes instead of 64, so that every wave gets a share of a
                // small level -- 48.6 ms against 47.8; the small levels are bound by their chain of round trips, not by idle waves.)
                constexpr uint32_t bsz = 64;
                if (L == 0) { drained = true; if (me == src_owner && wid == 0) { if (lane == 0) list[0] = (uint16_t)src_local; npend = 1; } }
                for (;;) {
                    while (npend < bsz) { // look at groups until a batch is full (the list holds 128)
                        if (gcur == gend) {
                            if (drained) break;
                            uint32_t g4 = 0;
                            if (lane == 0) g4 = atomicAdd(&s_gnext, (uint32_t)FORA_TEAM_DRAW);
                            g4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g4);
                            if (g4 >= ngroups) { drained = true; break; }
                            gcur = g4; gend = min(g4 + (uint32_t)FORA_TEAM_DRAW, ngroups);
                        }
                        const uint32_t gi = gcur++;
                        const unsigned long long mk = s_gmask[gi];
                        if ((mk >> lane) & 1ull) list[npend + (uint32_t)rank_below(mk)] = (uint16_t)(gi * 64 + lane);
                        npend += (uint32_t)__popcll(mk);
                    }
                    if (npend == 0) break;
                    // ---- pop up to 64 nodes, one per lane (algo.h:983-1002)
                    const uint32_t m = min(npend, 64u);
                    uint32_t cnt = 0, ebeg = 0;
                    uint32_t ebase = 0; // my entries of the level's increment table: ebase + lane
                    if (lane == 0) ebase = atomicAdd(&s_ncross, m);
                    ebase = (uint32_t)__builtin_amdgcn_readfirstlane((int)ebase);
                    if ((uint32_t)lane < m) {
                        const uint32_t l = list[lane];
                        const bool spare = l == R; // the source without in-edges: no local tables
                        const uint64_t rw = spare ? 0ull : rowl[l];
                        const uint32_t logi = logbase + ebase + (uint32_t)lane;     // my entry of the reserve log
                        const bool direct = spare || logi >= rlog_cap;             // (rare) straight to the accumulator
                        uint64_t rsv_old = 0;
                        uint32_t srcq = 0;
                        if (direct) { // (its operands are read here: they do not ride through the level in registers)
                            const TeamArgs ar = team_args();
                            rsv_old = spare ? ar->ppr[slab + src] : (ar->rsvl + ((uint64_t)team * T + me) * R)[l];
                            if (spare) srcq = ar->rowq[src];
                        }
                        const uint64_t rr = res[l];
                        res[l] = 0;                                       // algo.h:984-985
                        uint32_t deg = spare ? src_deg : (uint32_t)(rw >> 19) & 8191u;
                        if (!spare && deg == 8191u) deg = team_args()->deg[(uint32_t)rw & 0x7FFFFu]; // a hub: its exact degree
                        uint64_t rsv_add, dang;
                        const uint64_t inc = pop_value(afix, rr, deg, rsv_add, dang);
                        if (logi < rlog_cap) { rlog_val[logi] = direct ? 0ull : rsv_add; rlog_id[logi] = direct ? (uint16_t)0 : (uint16_t)l; } // algo.h:986-989, see TeamDev::rlog_id (the spare id's pop leaves an empty entry)
                        if (direct && rsv_add) { // (this member owns the node)
                            const TeamArgs ar = team_args();
                            if (spare) ar->ppr[slab + src] = rsv_old + rsv_add;
                            else { (ar->rsvl + ((uint64_t)team * T + me) * R)[l] = rsv_old + rsv_add; s_rsvovf = 1u; }
                        }
                        acc_res += rsv_add; my_dang += dang; acc_pops++; acc_relax += deg;
                        ebeg = spare ? srcq : (uint32_t)(rw >> 32); // (first quad of the row)
                        cnt = inc ? deg : 0u;
                        if (cnt) tout[ebase + lane] = inc;
                        winc[lane] = inc;
                        if (cnt > TEAM_HEAVY) { // a hub's row: cut into chunks that the waves take as they run out of rows of their own
                            const uint32_t hi = atomicAdd(&s_nheavy, 1u);
                            if (hi < (uint32_t)TEAM_NHEAVY) {
                                h_ent[hi] = ebase + lane; h_ebeg[hi] = ebeg; h_inc[hi] = inc;
                                h_cstart[hi] = atomicAdd(&s_hchunks, (cnt + TEAM_CHUNK - 1) / TEAM_CHUNK);
                                __atomic_store_n(&h_deg[hi], cnt, __ATOMIC_RELEASE); // (LDS, this wave's writes in order: the row is complete when its degree shows)
                                cnt = 0;
                            }
                        }
                    }
                    if (npend > 64) { // keep the entries behind the first 64
                        const uint16_t keep = list[64 + (lane < 63 ? lane : 63)];
                        __builtin_amdgcn_wave_barrier();
                        if ((uint32_t)lane < npend - 64) list[lane] = keep;
                    }
                    npend -= m;
                    uint32_t total;
                    const uint32_t qn = (cnt + 3u) >> 2;           // quads of this lane's row
                    const uint32_t pre = wave_excl_scan(qn, total);
                    TSTAMP(2);
                    // ---- their out-edges: a lane takes ONE QUAD (four consecutive edges of one row: one 16-byte load) of the
                    // concatenated rows.  Which row a quad belongs to comes from MARKS: every row with edges writes its number at the
                    // chunk position of its first quad, a lane reads the mark of its quad, and a wave scan ("the last mark so far": marks
                    // grow with the lane, so a prefix maximum) fills the gaps.  (Round 4 read four single edges per lane: four row
                    // look-ups, four exchanges and four loads per lane and chunk; a vector memory instruction costs the CU's address
                    // unit its 16+ cycles whatever it carries -- profiles/r05_team_probes.txt.)
                    static_assert(TEAM_EPT == 4, "a quad is four edges");
                    const uint32_t rowbase = ebeg - pre; // quad q of the concatenation, in this lane's row: colt quad rowbase + q (mod 2^32)
                    uint32_t carry = 0;                  // row (+ 1) of the last quad of the chunks before
                    const uint4 *colt4 = (const uint4 *)colt;
                    for (uint32_t cb = 0; cb < total; cb += 64) {
                        if (lane < 16) ((uint32_t *)mark)[lane] = 0;
                        __builtin_amdgcn_wave_barrier();
                        if (qn && pre - cb < 64u) mark[pre - cb] = (uint8_t)(lane + 1); // (pre >= cb: rows before were marked in their chunk)
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t mk = mark[lane];
                        const uint32_t x = wave_incl_scan_max(mk);
                        const uint32_t lastx = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
                        const uint32_t row1 = x ? x : carry; // the last mark at or before this lane's quad
                        if (lastx) carry = lastx;
                        const uint32_t si = row1 ? row1 - 1u : 0u;
                        const uint32_t q = cb + lane;
                        const uint32_t rb = (uint32_t)__shfl((int)rowbase, (int)si);
                        uint4 w4 = colt4[q < total ? rb + q : 0u]; // (every lane takes part in the exchange; the load without a branch around it)
                        if (q >= total) w4 = make_uint4(TEAM_EMPTY, TEAM_EMPTY, TEAM_EMPTY, TEAM_EMPTY);
                        const uint32_t w[TEAM_EPT] = {w4.x, w4.y, w4.z, w4.w};
                        const uint32_t entw = (ebase + si) << TEAM_LBITS;
                        uint32_t word[TEAM_EPT], dst[TEAM_EPT];
#pragma unroll
                        for (int k = 0; k < TEAM_EPT; k++) {
                            dst[k] = w[k] == TEAM_EMPTY ? TEAM_EMPTY : w[k] >> TEAM_LBITS;
                            word[k] = (w[k] & TEAM_LMASK) | entw;
                            if (w[k] != TEAM_EMPTY && (w[k] & 0x80000000u)) { // a hub: summed here
                                atomicAdd(&s_hub[w[k] & 0x7FFFFFFFu], (unsigned long long)winc[si]);
                                dst[k] = TEAM_EMPTY;
                            }
                        }
                        team_emit(word, dst, s_fill, mout);
                    }
                    TSTAMP(3);
                }
            }
            TSTAMP(2);
            // the dangling nodes' mass returns to the source within the level (algo.h:993-998): one message to its owner
            my_dang = wave_sum(my_dang);
            if (lane == 0 && my_dang) atomicAdd(&s_dang, (unsigned long long)my_dang);
            { // heavy rows, chunk by chunk, until every wave is through its own rows and no chunk is left: consecutive lanes on
              // consecutive edges.  (First form: all heavy rows after a barrier -- the waves that had finished early waited there.)
                if (lane == 0) atomicAdd(&s_wdone, 1u);
                for (;;) {
                    uint32_t k = TEAM_EMPTY, fin = 0;
                    if (lane == 0) {
                        const uint32_t nx = __atomic_load_n(&s_hnext, __ATOMIC_RELAXED), tot = __atomic_load_n(&s_hchunks, __ATOMIC_RELAXED);
                        if (nx < tot) { if (atomicCAS(&s_hnext, nx, nx + 1u) == nx) k = nx; }
                        else if (__atomic_load_n(&s_wdone, __ATOMIC_RELAXED) == (uint32_t)TEAM_NW && nx >= __atomic_load_n(&s_hchunks, __ATOMIC_RELAXED)) fin = 1;
                    }
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                    fin = (uint32_t)__builtin_amdgcn_readfirstlane((int)fin);
                    if (fin) break;
                    if (k == TEAM_EMPTY) { TSTAMP_SLEEP(9, 1, 30); continue; }
                    const TeamArgs a = team_args(); // (a heavy chunk is 256 edges: two scalar loads are nothing beside it)
                    uint32_t *mout = a->msg + ((uint64_t)team * 2 + (g & 1u)) * cap_total;
                    const uint32_t *colt = a->colt;
                    uint32_t eb = 0, dgh = 0, ent = 0, c0 = 0;
                    uint64_t hinc = 0;
                    for (bool found = false; !found;) { // the row of chunk k (it may still be on its way into the list): lane h looks at entries h, h + 64
                        const uint32_t nh = min(__atomic_load_n(&s_nheavy, __ATOMIC_RELAXED), (uint32_t)TEAM_NHEAVY);
                        for (uint32_t hb = 0; hb < nh && !found; hb += 64) {
                            const uint32_t h = hb + lane;
                            const uint32_t dg = h < nh ? __atomic_load_n(&h_deg[h], __ATOMIC_ACQUIRE) : 0u;
                            const uint32_t cs = dg ? h_cstart[h] : 0u;
                            const unsigned long long hit = __ballot(dg && k >= cs && k < cs + (dg + TEAM_CHUNK - 1) / TEAM_CHUNK);
                            if (hit) {
                                const uint32_t hs = hb + (uint32_t)__ffsll((long long)hit) - 1u;
                                eb = h_ebeg[hs]; dgh = h_deg[hs]; ent = h_ent[hs] << TEAM_LBITS; hinc = h_inc[hs]; c0 = (k - h_cstart[hs]) * TEAM_CHUNK;
                                found = true;
                            }
                        }
                    }
                    uint32_t word[TEAM_EPT], dst[TEAM_EPT];
                    {
                        const uint32_t q = (c0 >> 2) + lane; // my quad of the row (consecutive lanes on consecutive quads: 1 KB per load)
                        uint4 w4 = make_uint4(TEAM_EMPTY, TEAM_EMPTY, TEAM_EMPTY, TEAM_EMPTY);
                        if (q < ((dgh + 3u) >> 2)) w4 = ((const uint4 *)colt)[(uint64_t)eb + q];
                        const uint32_t w[TEAM_EPT] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                        for (int kk = 0; kk < TEAM_EPT; kk++) {
                            dst[kk] = TEAM_EMPTY; word[kk] = 0;
                            if (w[kk] != TEAM_EMPTY) {
                                if (w[kk] & 0x80000000u) atomicAdd(&s_hub[w[kk] & 0x7FFFFFFFu], (unsigned long long)hinc);
                                else { dst[kk] = w[kk] >> TEAM_LBITS; word[kk] = (w[kk] & TEAM_LMASK) | ent; }
                            }
                        }
                    }
                    team_emit(word, dst, s_fill, mout);
                }
            }
            TSTAMP(6);
            __syncthreads(); // (the message stores are waited for below, together with the counts)
            TSTAMP(7);
            {
                const TeamArgs a = team_args();
                uint32_t *mout = a->msg + ((uint64_t)team * 2 + (g & 1u)) * cap_total;
                uint64_t *tout = a->inct + (((uint64_t)team * 2 + (g & 1u)) * T + me) * tstride;
                if (tid == 0 && s_dang) { // one more table entry, one more message
                    const uint32_t ent = s_ncross, pos = atomicAdd(&s_fill[src_owner], 1u);
                    tout[ent] = s_dang;
                    mout[pos] = src_local | (ent << TEAM_LBITS);
                }
                if ((uint32_t)tid < H) { // (wave-uniform up to the last wave) the hubs' sums of this level: one message each
                    const uint32_t *hubtgt = a->hubtgt;
                    for (uint32_t h = tid; h < H; h += TEAM_THREADS) {
                        const unsigned long long hv = s_hub[h];
                        if (hv) {
                            s_hub[h] = 0;
                            const uint32_t tg = h < (uint32_t)TEAM_THREADS ? hub_tg0 : hubtgt[h], ent = s_ncross + 1 + atomicAdd(&s_hubent, 1u);
                            tout[ent] = hv;
                            mout[atomicAdd(&s_fill[tg >> TEAM_LBITS], 1u)] = (tg & TEAM_LMASK) | (ent << TEAM_LBITS);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave: its message and table stores have completed
                __syncthreads();                                 // (and s_fill is final)
                // ================= my words of the level (see the consume): the other members take it from here
                const uint32_t tag = (g + 1u) & 0xFFFFFFu;
                if (!same_xcd) {
                    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    __syncthreads();
                }
                if ((uint32_t)tid < T) {
                    unsigned long long *cw = team_args()->cntw + ((uint64_t)team * 2 + (g & 1u)) * T * T; // [destination][source]
                    __hip_atomic_store(&cw[(uint32_t)tid * T + me], (unsigned long long)(s_fill[tid] - s_moff[tid]) | ((unsigned long long)s_ncross << 24) | ((unsigned long long)tag << 40),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                logbase += s_ncross; // (s_fill / s_ncross are zeroed in the sweep of the next level: behind the barrier that closes its consume)
            }
            TSTAMP(5);
            TSTAMP_LEVEL(L);
            g++;
        }
        // ---- the slot's counters (algo.h:992 rsum bookkeeping)
        acc_res = wave_sum(acc_res); acc_pops = wave_sum(acc_pops); acc_relax = wave_sum(acc_relax);
        const int tid = fresh(tid0), lane = tid & 63;
        if (tid < 3) s_acc[tid] = 0;
        __syncthreads();
        if (lane == 0 && acc_pops) {
            atomicAdd(&s_acc[0], (unsigned long long)acc_res);
            atomicAdd(&s_acc[1], (unsigned long long)acc_pops);
            atomicAdd(&s_acc[2], (unsigned long long)acc_relax);
        }
        __syncthreads();
        if (tid == 0) {
            QState *s = &team_args()->qs[q];
            if (s_acc[1]) {
                atomicAdd(&s->reserved, s_acc[0]);
                atomicAdd(&s->pops, s_acc[1]);
                if (s_acc[2]) atomicAdd(&s->relax, s_acc[2]);
            }
            if (me == 0 && nlev) atomicAdd(&s->levels, nlev);
        }
    }
    TSTAMP_FLUSH();
}

} // namespace fora
