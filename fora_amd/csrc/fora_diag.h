// fora_diag.h -- every diagnostic hook of the hot kernels in one place.
//
// The product library (fora_amd/build.py: no -DFORA_PROBE_* / -DFORA_STAMPS* / -DFORA_DG_FAKE_*) compiles every hook
// below to nothing: the kernel bodies in fora_kernels.h / fora_team.h carry no #ifdef of their own, they call these
// hooks.  Any of the macros makes FORA_DIAG_BUILD 1; fora_hip_get_option(NULL, "diag_build") reports it and
// tests/test_capi_cpu.py asserts 0 on the shipped fora_amd/libfora_hip.so, so a probe build (some of them compute
// WRONG results on purpose: FORA_DG_FAKE_*) cannot ship by a -D typo.
//
//   stamps   FORA_STAMPS            cycles per phase of thread 0, summed over workgroups (fora_hip_get_stamps)
//            FORA_STAMPS_SMALL=N    team push: only the levels whose predecessor popped <= N nodes of the slot
//            FORA_STAMPS_LEVELS     team push: cycles per level number
//   probes   "one more access per message / edge, results unchanged": what does the access cost where it stands?
//            FORA_PROBE_STORE=off / FORA_PROBE_STORE2=1|2 / FORA_PROBE_GATHER        k_push_team (profiles/r05_team_probes.txt)
//            FORA_PROBE_BIN_LOAD / _RANK / _STORE / _SYNC / _LDS                     k_pushq_bin  (profiles/r06_wide_probes.txt)
//            FORA_PROBE_ACC_LOAD / _ATOM / _SWEEP                                    k_accum<false>
//   fakes    FORA_DG_FAKE_STEP0 / FORA_DG_FAKE_ALL   k_walk_dg with its gathers bent into a 4-KB window: WRONG results, timing floor only
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(FORA_STAMPS) || defined(FORA_STAMPS_SMALL) || defined(FORA_STAMPS_LEVELS) || defined(FORA_PROBE_STORE) || defined(FORA_PROBE_STORE2) || \
    defined(FORA_PROBE_GATHER) || defined(FORA_PROBE_BIN_LOAD) || defined(FORA_PROBE_BIN_RANK) || defined(FORA_PROBE_BIN_STORE) ||              \
    defined(FORA_PROBE_BIN_SYNC) || defined(FORA_PROBE_BIN_LDS) || defined(FORA_PROBE_ACC_LOAD) || defined(FORA_PROBE_ACC_ATOM) ||               \
    defined(FORA_PROBE_ACC_SWEEP) || defined(FORA_DG_FAKE_STEP0) || defined(FORA_DG_FAKE_ALL)
#define FORA_DIAG_BUILD 1
#else
#define FORA_DIAG_BUILD 0
#endif

// ------------------------------------------------------------------ phase stamps (macros: they declare / use locals of the kernel)
#ifdef FORA_STAMPS
// bucketed kernels: Dev::stamps [0..15] bin kernel, [16..31] accumulate
#define STAMP_DECL long long st_t_ = clock64(); unsigned long long st_a_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)st_a_;
#define STAMP(slot) do { const long long n_ = clock64(); st_a_[(slot) & 7] += (unsigned long long)(n_ - st_t_); st_t_ = n_; } while (0)
#define STAMP_FLUSH(base) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; i_++) if (st_a_[i_]) atomicAdd(&d.stamps[(base) + i_], st_a_[i_]); } while (0)
// team push: TeamDev::stamps [0..9]
#define TSTAMP_DECL long long ts_t_ = clock64(); unsigned long long ts_a_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; bool ts_on_ = true; (void)ts_on_;
#ifdef FORA_STAMPS_SMALL
#define TSTAMP(k) do { const long long n_ = clock64(); if (ts_on_) ts_a_[k] += (unsigned long long)(n_ - ts_t_); ts_t_ = n_; } while (0)
#define TSTAMP_GATE(F) do { ts_on_ = (F) <= (uint32_t)(FORA_STAMPS_SMALL) && (F) > 0; if (ts_on_) ts_a_[4]++; } while (0) // (levels counted in slot 4)
#else
#define TSTAMP(k) do { const long long n_ = clock64(); ts_a_[k] += (unsigned long long)(n_ - ts_t_); ts_t_ = n_; } while (0)
#define TSTAMP_GATE(F) do {} while (0)
#endif
#define TSTAMP_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 10; i_++) if (ts_a_[i_]) atomicAdd(&team_args()->stamps[i_], ts_a_[i_]); } while (0)
// s_sleep(n) of a poll loop, charged to slot k as pure waiting (+ `extra` cycles for the poll itself)
#define TSTAMP_SLEEP(k, n, extra) do { const long long w0_ = clock64(); __builtin_amdgcn_s_sleep(n); ts_a_[k] += (unsigned long long)(clock64() - w0_) + (extra); } while (0)
#else
#define STAMP_DECL
#define STAMP(slot) do {} while (0)
#define STAMP_FLUSH(base) do {} while (0)
#define TSTAMP_DECL
#define TSTAMP(k) do {} while (0)
#define TSTAMP_GATE(F) do {} while (0)
#define TSTAMP_FLUSH() do {} while (0)
#define TSTAMP_SLEEP(k, n, extra) __builtin_amdgcn_s_sleep(n)
#endif
#ifdef FORA_STAMPS_LEVELS
#define TSTAMP_LEVEL_DECL long long lv_t_ = clock64();
#define TSTAMP_LEVEL(L) do { if (threadIdx.x == 0) { const long long n_ = clock64(); atomicAdd(&team_args()->stamps[(L) < 31 ? (L) : 31], (unsigned long long)(n_ - lv_t_)); lv_t_ = n_; } } while (0)
#else
#define TSTAMP_LEVEL_DECL
#define TSTAMP_LEVEL(L) do {} while (0)
#endif

namespace fora {
namespace diag {

// a value the optimiser must keep: the probe's load has a consumer, its result changes nothing
__device__ __forceinline__ void keep(uint32_t x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void keep(uint64_t x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void keep(const uint4 &x) { asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)); }
// a zero the optimiser cannot see through (an LDS atomic that adds it is a real atomic)
__device__ __forceinline__ unsigned long long opaque_zero() { unsigned long long z = 0; asm volatile("" : "+v"(z)); return z; }

// words the team push needs behind its message buffer for the store probes (fora_hip.hip: ensure_workspace)
#if defined(FORA_PROBE_STORE) || defined(FORA_PROBE_STORE2)
constexpr size_t TEAM_MSG_PROBE_WORDS = (size_t)80000000 + (1u << 18);
#else
constexpr size_t TEAM_MSG_PROBE_WORDS = 0;
#endif

// ---- k_push_team (fora_team.h).  EPT messages of a chunk: word[k] stored at mout[slot[k]] when dst[k] != empty.
template <int EPT>
__device__ __forceinline__ void team_emit_probe(const uint32_t (&word)[EPT], const uint32_t (&dst)[EPT], const uint32_t (&slot)[EPT], uint32_t *mout, uint32_t empty) {
#ifdef FORA_PROBE_STORE2 // one more store per message whose lanes write consecutive words (1) / four runs of 16 words (2)
    {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot[0]) & ~63u, lane_ = threadIdx.x & 63u;
#pragma unroll
        for (int k = 0; k < EPT; k++)
            mout[(size_t)80000000 + base + k * 4096 + (FORA_PROBE_STORE2 == 1 ? lane_ : (lane_ >> 4) * 1024 + (lane_ & 15u) + 5)] = word[k];
    }
#endif
#ifdef FORA_PROBE_STORE // ONE MORE scattered 4-byte store per message (a second copy, FORA_PROBE_STORE words further on)
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (dst[k] != empty) mout[(size_t)(FORA_PROBE_STORE) + slot[k]] = word[k];
#endif
    (void)word; (void)dst; (void)slot; (void)mout; (void)empty;
}
// ONE MORE scattered 8-byte gather per consumed message: another line of the same increment table; adds 0 (values are below 2^62)
__device__ __forceinline__ uint64_t team_gather_probe(const uint64_t *tb, uint32_t entry, bool have) {
#ifdef FORA_PROBE_GATHER
    const uint64_t xa = __hip_atomic_load((const unsigned long long *)&tb[have ? (entry ^ 8u) : 0u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return xa >> 63;
#else
    (void)tb; (void)entry; (void)have;
    return 0;
#endif
}

// ---- k_walk_dg (fora_kernels.h): the copy id a move starts from.  The fakes bend it into a 4-KB window (WRONG results)
__device__ __forceinline__ uint32_t dg_from(uint32_t cur, uint32_t t) {
#if defined(FORA_DG_FAKE_ALL)
    (void)t; return cur & 1023u;
#elif defined(FORA_DG_FAKE_STEP0)
    return t == 0 ? (cur & 1023u) : cur;
#else
    (void)t; return cur;
#endif
}

// ---- k_pushq_bin (wide layouts; profiles/r06_wide_probes.txt)
// one more 16-byte load per quad: the same quad of the OTHER quad copy of col (col4 / col_hub4: real HBM traffic) or, when the
// graph has only one, of the same array again through a laundered pointer (address path only).  The value is kept alive by
// bin_load_keep() where the real quads are unpacked, so the probe's load is in flight together with them.
__device__ __forceinline__ uint4 bin_load_probe(const int32_t *colsrc, const int32_t *col4, const int32_t *col_hub4, const int32_t *col, int64_t quad) {
#if defined(FORA_PROBE_BIN_LOAD) && FORA_PROBE_BIN_LOAD == 2 // a line of the UNPADDED copy of col (other HBM lines, same access pattern): index scaled to stay inside it
    (void)colsrc; (void)col4; (void)col_hub4;
    const uint4 *pp = (const uint4 *)col + (quad - (quad >> 3));
    asm volatile("" : "+v"(pp));
    return *pp;
#elif defined(FORA_PROBE_BIN_LOAD)
    (void)col;
    const int32_t *other = (col4 && col_hub4) ? (colsrc == col4 ? col_hub4 : col4) : colsrc;
    const uint4 *pp = (const uint4 *)other + quad;
    asm volatile("" : "+v"(pp));
    return *pp;
#else
    (void)colsrc; (void)col4; (void)col_hub4; (void)col; (void)quad;
    return make_uint4(0u, 0u, 0u, 0u);
#endif
}
__device__ __forceinline__ void bin_load_keep(const uint4 &p) {
#ifdef FORA_PROBE_BIN_LOAD
    keep(p);
#else
    (void)p;
#endif
}
#ifdef FORA_PROBE_BIN_RANK
#define BIN_RANK_PROBE_DECL(NB) __shared__ uint32_t s_cnt_probe[NB];
#define BIN_RANK_PROBE(b) fora::diag::keep(atomicAdd(&s_cnt_probe[b], 1u)) // one more returning LDS add per message (a mirror histogram)
#else
#define BIN_RANK_PROBE_DECL(NB)
#define BIN_RANK_PROBE(b) do {} while (0)
#endif
// one more 8-byte store per message: the same word to the same place (one more request through the address path and L2; the line is written once)
__device__ __forceinline__ void bin_store_probe(uint64_t *at, uint64_t word) {
#ifdef FORA_PROBE_BIN_STORE
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(at), "v"(word) : "memory");
#else
    (void)at; (void)word;
#endif
}
__device__ __forceinline__ void bin_sync_probe() { // two more workgroup barriers per chunk
#ifdef FORA_PROBE_BIN_SYNC
    __syncthreads(); __syncthreads();
#endif
}
// one more LDS write and one more LDS read per message of the stage
__device__ __forceinline__ void bin_lds_write_probe(uint32_t *s_msg, uint32_t sp, uint32_t v) {
#ifdef FORA_PROBE_BIN_LDS
    *(volatile uint32_t *)&s_msg[sp] = v;
#else
    (void)s_msg; (void)sp; (void)v;
#endif
}
__device__ __forceinline__ void bin_lds_read_probe(const uint32_t *s_msg, uint32_t m) {
#ifdef FORA_PROBE_BIN_LDS
    keep(*(const volatile uint32_t *)&s_msg[m]);
#else
    (void)s_msg; (void)m;
#endif
}

// ---- k_accum<false> (wide layouts)
__device__ __forceinline__ uint64_t acc_load_probe(const uint64_t *p) { // one more (plain) load of the message word: through L1 / L2
#ifdef FORA_PROBE_ACC_LOAD
    asm volatile("" : "+v"(p));
    return *p;
#else
    (void)p; return 0;
#endif
}
__device__ __forceinline__ void acc_load_keep(uint64_t v) {
#ifdef FORA_PROBE_ACC_LOAD
    keep(v);
#else
    (void)v;
#endif
}
__device__ __forceinline__ void acc_atom_probe(uint64_t *acc_word) { // one more LDS add per message (adds an opaque zero)
#ifdef FORA_PROBE_ACC_ATOM
    atomicAdd((unsigned long long *)acc_word, opaque_zero());
#else
    (void)acc_word;
#endif
}
// the sweep's residue and degree loads once more (laundered pointers: real second loads, in flight with the first ones)
__device__ __forceinline__ uint64_t acc_sweep_probe(const uint64_t *res_word, const uint32_t *deg_word) {
#ifdef FORA_PROBE_ACC_SWEEP
    asm volatile("" : "+v"(res_word));
    asm volatile("" : "+v"(deg_word));
    return *res_word + *deg_word;
#else
    (void)res_word; (void)deg_word; return 0;
#endif
}
__device__ __forceinline__ void acc_sweep_keep(uint64_t v) {
#ifdef FORA_PROBE_ACC_SWEEP
    keep(v);
#else
    (void)v;
#endif
}

} // namespace diag
} // namespace fora
