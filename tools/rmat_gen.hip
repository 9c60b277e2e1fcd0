// rmat_gen.hip -- TEST / BENCH TOOLING, not part of the product library.
//
// Deterministic R-MAT graph generator on the GPU (the real edge lists of the BASELINE
// configs are not available, every measured graph is an R-MAT graph with the real graph's n and m).
// The numpy generator of fora_amd/synth.py needs ~13 minutes for a Twitter-2010-sized graph; this one
// builds it in seconds, so that LiveJournal- and Twitter-sized cases fit in driver-run tests and bench.py.
//
//   rmat_generate(n, m, seed, plain_rmat, row_ptr[n+1], col[m])
//     directed R-MAT (a, b, c, d) = (0.57, 0.19, 0.19, 0.05) over ceil(log2 n) levels, ids >= n rejected,
//     node ids randomly permuted, self loops dropped, duplicates removed, EXACTLY m distinct edges
//     (surplus candidates are dropped by a hash of the edge, so the cut is unbiased);
//     plain_rmat = 0: every node first gets one uniform random out-edge (real web graphs have almost no
//     zero-out-degree nodes), plain_rmat = 1: plain R-MAT (about 43 % dangling nodes at ws size).
//     Output: CSR on the host, rows sorted by target.  Same (n, m, seed, mode) -> same graph, bit for bit:
//     counter-based hashing, integer arithmetic, radix sort.
#include <hip/hip_runtime.h>
#include <cstring>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <stdint.h>
#include <stdio.h>
#include <chrono>

namespace {

constexpr int TB = 256;
constexpr uint64_t NONE = ~0ull;

__host__ __device__ inline uint64_t mix64(uint64_t x) { // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_perm_keys(int64_t n, uint64_t seed, uint64_t *keys, uint32_t *vals) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    keys[i] = mix64(seed ^ mix64((uint64_t)i + 0x1111));
    vals[i] = (uint32_t)i;
}

// every node u gets one out-edge to a uniform v != u; tagged key = (u * n + v) << 1 (low bit 0: must be kept)
__global__ void k_base_edges(int64_t n, uint64_t seed, uint64_t *out) {
    const int64_t u = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (u >= n) return;
    uint64_t v = mix64(seed ^ mix64((uint64_t)u + 0x2222)) % (uint64_t)(n - 1);
    v += v >= (uint64_t)u;
    out[u] = ((uint64_t)u * (uint64_t)n + v) << 1;
}

// candidate c: one R-MAT edge; tagged key = (perm[s] * n + perm[d]) << 1 | 1, NONE when rejected
__global__ void k_rmat_edges(int64_t n, int scale, uint64_t seed, uint64_t c0, int64_t count, const uint32_t *perm,
                             uint64_t *out) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= count) return;
    const uint64_t c = c0 + (uint64_t)i;
    uint64_t s = 0, d = 0, bits = 0;
    int have = 0;
    uint64_t ctr = 0;
    // 16 bits per level: a = 0.57, a + b = 0.76, a + b + c = 0.95 of 65536
    for (int l = 0; l < scale; l++) {
        if (!have) { bits = mix64(seed ^ mix64(c * 8 + ctr + 0x3333)); ctr++; have = 4; }
        const uint32_t r = (uint32_t)bits & 0xFFFFu;
        bits >>= 16; have--;
        const uint32_t sb = r >= 49807u;                          // rows c, d
        const uint32_t db = (r >= 37356u && r < 49807u) || r >= 62259u; // columns b, d
        s = (s << 1) | sb;
        d = (d << 1) | db;
    }
    uint64_t k = NONE;
    if (s < (uint64_t)n && d < (uint64_t)n && s != d) k = (((uint64_t)perm[s] * (uint64_t)n + (uint64_t)perm[d]) << 1) | 1ull;
    out[i] = k;
}

__global__ void k_flag_first(const uint64_t *keys, int64_t count, uint8_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= count) return;
    const uint64_t k = keys[i];
    flag[i] = k != NONE && (i == 0 || (k >> 1) != (keys[i - 1] >> 1));
}

// selection hash: base edges 0 (always kept), candidates a 64-bit hash of the edge
__device__ inline uint64_t sel_hash(uint64_t tagged, uint64_t seed) { return (tagged & 1ull) ? (mix64(seed ^ mix64((tagged >> 1) + 0x4444)) | 1ull) : 0ull; }

__global__ void k_count_le(const uint64_t *keys, int64_t count, uint64_t seed, uint64_t T, unsigned long long *out) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < count; i += (int64_t)gridDim.x * TB)
        acc += sel_hash(keys[i], seed) <= T;
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

__global__ void k_flag_le(const uint64_t *keys, int64_t count, uint64_t seed, uint64_t T, uint8_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= count) return;
    flag[i] = sel_hash(keys[i], seed) <= T;
}

__global__ void k_split(const uint64_t *keys, int64_t count, int64_t n, int32_t *col, unsigned long long *deg) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= count) return;
    const uint64_t k = keys[i] >> 1;
    const uint64_t u = k / (uint64_t)n;
    col[i] = (int32_t)(k - u * (uint64_t)n);
    atomicAdd(&deg[u], 1ull);
}

struct Bufs {
    void *p[16];
    int np = 0;
    template <typename T> hipError_t alloc(T **out, size_t bytes) {
        hipError_t e = hipMalloc((void **)out, bytes ? bytes : 1);
        if (e == hipSuccess) p[np++] = *out;
        return e;
    }
    ~Bufs() { for (int i = 0; i < np; i++) (void)hipFree(p[i]); }
};

#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "rmat_gen: %s: %s\n", #call, hipGetErrorString(e_)); return -2; } } while (0)

unsigned blocks(int64_t n) { return (unsigned)((n + TB - 1) / TB); }

} // namespace

extern "C" int rmat_generate(int64_t n, int64_t m, uint64_t seed, int plain_rmat, int64_t *row_ptr, int32_t *col, double *seconds) {
    if (n < 2 || n >= (1ll << 31) || m < 0 || !row_ptr || (m && !col)) return -1;
    if (!plain_rmat && m < n) return -1;
    if ((double)m > (double)n * (double)(n - 1) * 0.5) return -1; // too dense for rejection sampling to finish
    const auto t0 = std::chrono::steady_clock::now();
    int scale = 1;
    while ((1ll << scale) < n) scale++;
    Bufs B;
    // ---- random permutation of the ids
    uint64_t *pk = nullptr, *pk2 = nullptr;
    uint32_t *pv = nullptr, *perm = nullptr;
    CHK(B.alloc(&pk, (size_t)n * 8)); CHK(B.alloc(&pk2, (size_t)n * 8));
    CHK(B.alloc(&pv, (size_t)n * 4)); CHK(B.alloc(&perm, (size_t)n * 4));
    hipLaunchKernelGGL(k_perm_keys, dim3(blocks(n)), dim3(TB), 0, 0, n, seed, pk, pv);
    size_t tmp_bytes = 0;
    void *tmp = nullptr;
    CHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, pk, pk2, pv, perm, (size_t)n));
    // ---- key buffers: unique tagged keys so far + a batch of new candidates, double-buffered for the sort
    const int64_t base = plain_rmat ? 0 : n;
    const int64_t cap = (int64_t)((double)m * 1.6) + (1 << 20) + base;
    uint64_t *ka = nullptr, *kb = nullptr;
    uint8_t *flag = nullptr;
    unsigned long long *d_cnt = nullptr;
    CHK(B.alloc(&ka, (size_t)cap * 8)); CHK(B.alloc(&kb, (size_t)cap * 8));
    CHK(B.alloc(&flag, (size_t)cap));
    CHK(B.alloc(&d_cnt, 16));
    size_t t2 = 0, t3 = 0;
    CHK(rocprim::radix_sort_keys(nullptr, t2, ka, kb, (size_t)cap));
    CHK(rocprim::select(nullptr, t3, ka, flag, kb, (size_t *)d_cnt, (size_t)cap));
    tmp_bytes = std::max(tmp_bytes, std::max(t2, t3));
    CHK(B.alloc((char **)&tmp, tmp_bytes));
    CHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, pk, pk2, pv, perm, (size_t)n));
    int64_t have = 0; // unique tagged keys in ka[0 .. have)
    if (base) { hipLaunchKernelGGL(k_base_edges, dim3(blocks(n)), dim3(TB), 0, 0, n, seed, ka); have = n; }
    uint64_t next_c = 0;
    for (int round = 0; have < m || round == 0; round++) {
        if (round > 64) return -3;
        const int64_t need = m - have;
        int64_t batch = (int64_t)((double)std::max<int64_t>(need, 0) * (round == 0 ? 1.35 : 1.6)) + (1 << 16);
        if (have + batch > cap) batch = cap - have;
        if (batch <= 0) return -3;
        hipLaunchKernelGGL(k_rmat_edges, dim3(blocks(batch)), dim3(TB), 0, 0, n, scale, seed, next_c, batch, (const uint32_t *)perm, ka + have);
        next_c += (uint64_t)batch;
        const int64_t tot = have + batch;
        size_t tb = tmp_bytes;
        CHK(rocprim::radix_sort_keys(tmp, tb, ka, kb, (size_t)tot)); // base edges (low bit 0) sort before their duplicates
        hipLaunchKernelGGL(k_flag_first, dim3(blocks(tot)), dim3(TB), 0, 0, (const uint64_t *)kb, tot, flag);
        tb = tmp_bytes;
        CHK(rocprim::select(tmp, tb, kb, flag, ka, (size_t *)d_cnt, (size_t)tot));
        size_t sel = 0;
        CHK(hipMemcpy(&sel, d_cnt, sizeof(size_t), hipMemcpyDeviceToHost));
        have = (int64_t)sel;
        // acceptance far below the estimate (tiny n): grow the next batch through `need`
    }
    // ---- cut to exactly m: keep the base edges and the candidates with the smallest hash (binary search on the threshold)
    int64_t fin = have;
    if (have > m) {
        uint64_t lo = 0, hi = NONE; // smallest T with count(h <= T) >= m
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            CHK(hipMemset(d_cnt, 0, 8));
            hipLaunchKernelGGL(k_count_le, dim3(4096), dim3(TB), 0, 0, (const uint64_t *)ka, have, seed, mid, d_cnt);
            unsigned long long c = 0;
            CHK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
            if ((int64_t)c >= m) hi = mid; else lo = mid + 1;
        }
        hipLaunchKernelGGL(k_flag_le, dim3(blocks(have)), dim3(TB), 0, 0, (const uint64_t *)ka, have, seed, lo, flag);
        size_t tb = tmp_bytes;
        CHK(rocprim::select(tmp, tb, ka, flag, kb, (size_t *)d_cnt, (size_t)have));
        size_t sel = 0;
        CHK(hipMemcpy(&sel, d_cnt, sizeof(size_t), hipMemcpyDeviceToHost));
        fin = (int64_t)sel;
        std::swap(ka, kb);
    }
    if (fin != m) { fprintf(stderr, "rmat_gen: selected %lld edges, wanted %lld (hash tie at the cut)\n", (long long)fin, (long long)m); return -4; }
    // ---- CSR
    int32_t *d_col = nullptr;
    unsigned long long *d_deg = nullptr, *d_rp = nullptr;
    CHK(B.alloc(&d_col, (size_t)m * 4));
    CHK(B.alloc(&d_deg, ((size_t)n + 1) * 8)); CHK(B.alloc(&d_rp, ((size_t)n + 1) * 8));
    CHK(hipMemset(d_deg, 0, ((size_t)n + 1) * 8));
    hipLaunchKernelGGL(k_split, dim3(blocks(m)), dim3(TB), 0, 0, (const uint64_t *)ka, m, n, d_col, d_deg);
    size_t t4 = 0;
    CHK(rocprim::exclusive_scan(nullptr, t4, d_deg, d_rp, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>()));
    if (t4 > tmp_bytes) return -5;
    CHK(rocprim::exclusive_scan(tmp, t4, d_deg, d_rp, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>()));
    CHK(hipMemcpy(row_ptr, d_rp, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
    if (m) CHK(hipMemcpy(col, d_col, (size_t)m * 4, hipMemcpyDeviceToHost));
    CHK(hipDeviceSynchronize());
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
