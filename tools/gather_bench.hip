// Experiment tooling: dependent random gathers, K independent chains per lane -- does the chip's random 8-byte load
// rate at a walk-like working set (6.6 MB) grow with the loads a lane keeps in flight, or is it a hard rate?
// usage: gather_bench [table MB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
template <int K>
__global__ void __launch_bounds__(256) k_chase(const uint64_t *tab, uint32_t mask, int steps, uint64_t *out) {
    uint32_t idx[K];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) idx[k] = (t * 2654435761u + k * 40503u) & mask;
    uint64_t acc = 0;
    for (int s = 0; s < steps; s++) {
        uint64_t v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = tab[idx[k]];
#pragma unroll
        for (int k = 0; k < K; k++) { idx[k] = (uint32_t)v[k] & mask; acc += v[k]; }
    }
    if (acc == 1) out[0] = acc;
}
template <int K> void run(const uint64_t *d_tab, uint32_t mask, uint64_t *d_out, int blocks_per_cu) {
    const int steps = 512 / K;
    const int blocks = 256 * blocks_per_cu * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_chase<K>, dim3(blocks), dim3(256), 0, 0, d_tab, mask, steps, d_out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_chase<K>, dim3(blocks), dim3(256), 0, 0, d_tab, mask, steps, d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double loads = (double)blocks * 256 * steps * K;
    printf("K=%d chains/lane, %d blocks of 256 (x4 per CU slot %d): %.1f G loads/s\n", K, blocks, blocks_per_cu, loads / ms / 1e6);
}
int main(int argc, char **argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 6.6;
    uint32_t n = 1; while ((double)n * 2 * 8 <= mb * 1048576.0) n *= 2; // power of two entries
    uint64_t *h = (uint64_t *)malloc((size_t)n * 8);
    uint64_t x = 88172645463325252ULL;
    for (uint32_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x; }
    uint64_t *d_tab, *d_out; hipMalloc(&d_tab, (size_t)n * 8); hipMalloc(&d_out, 8);
    hipMemcpy(d_tab, h, (size_t)n * 8, hipMemcpyHostToDevice);
    printf("table %.1f MB (%u entries)\n", n * 8.0 / 1048576, n);
    for (int bpc = 2; bpc <= 8; bpc += 3) { run<1>(d_tab, n - 1, d_out, bpc); run<2>(d_tab, n - 1, d_out, bpc); run<4>(d_tab, n - 1, d_out, bpc); run<8>(d_tab, n - 1, d_out, bpc); }
    return 0;
}
