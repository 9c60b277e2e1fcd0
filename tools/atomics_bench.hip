// Microbenchmark behind DESIGN.md's choice of accumulation primitive: scattered 8-byte
// accumulations into a table of `bytes` size, one address per lane, uniformly random.
//   hipcc --offload-arch=gfx950 -O3 -o atomics_bench tools/atomics_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void k(uint64_t *t, uint64_t mask, int iters, uint64_t *sink) {
    uint32_t s = mix(blockIdx.x * 256 + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        s = mix(s + i);
        uint64_t a = ((uint64_t)s * 2654435761ull >> 8) & mask;
        if (MODE == 0) acc += atomicAdd((unsigned long long *)&t[a], 3ull);                 // u64 returning
        if (MODE == 1) atomicAdd((unsigned long long *)&t[a], 3ull);                        // u64 no return
        if (MODE == 2) acc += (uint64_t)atomicAdd((double *)&t[a], 1.0);                    // f64 returning
        if (MODE == 3) atomicAdd((double *)&t[a], 1.0);                                     // f64 no return
        if (MODE == 4) acc += __hip_atomic_fetch_add((unsigned long long *)&t[a], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 5) { uint64_t v = t[a]; t[a] = v + 3; }                                 // racy RMW (upper bound)
        if (MODE == 6) acc += t[a];                                                         // random 8-B loads
        if (MODE == 7) acc += atomicAdd((unsigned int *)&t[a], 3u);                         // u32 returning
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <int MODE>
double run(uint64_t *t, uint64_t n, int iters, uint64_t *sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 4096;
    k<MODE><<<blocks, 256>>>(t, n - 1, 4, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(t, n - 1, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * iters / (ms * 1e-3) / 1e9; // G ops/s
}

int main() {
    uint64_t *sink; hipMalloc(&sink, 8);
    const char *names[] = {"u64 atomic ret", "u64 atomic noret", "f64 atomic ret", "f64 atomic noret", "u64 wg-scope ret", "racy load+store", "random 8B load", "u32 atomic ret"};
    for (uint64_t mb : {2ull, 16ull, 128ull, 1024ull, 8192ull}) {
        uint64_t n = mb * 1024 * 1024 / 8;
        uint64_t *t; if (hipMalloc(&t, n * 8) != hipSuccess) { printf("alloc fail\n"); return 1; }
        hipMemset(t, 0, n * 8);
        double r[8];
        r[0] = run<0>(t, n, 64, sink); r[1] = run<1>(t, n, 64, sink); r[2] = run<2>(t, n, 64, sink); r[3] = run<3>(t, n, 64, sink);
        r[4] = run<4>(t, n, 64, sink); r[5] = run<5>(t, n, 64, sink); r[6] = run<6>(t, n, 64, sink); r[7] = run<7>(t, n, 64, sink);
        printf("table %5llu MiB:", (unsigned long long)mb);
        for (int i = 0; i < 8; i++) printf("  %s %.1f G/s;", names[i], r[i]);
        printf("\n");
        hipFree(t);
    }
    return 0;
}
