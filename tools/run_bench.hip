// Experiment tooling: what does the chip deliver for traffic that comes in RUNS of R bytes at scattered places?
// The push kernels write messages in runs of 30-230 bytes, pop rows of 30-50 bytes, and read-modify-write 8-byte
// residues; DESIGN.md section 9 prices them against this table rather than against the 8 TB/s streaming peak.
// Consecutive lanes touch consecutive 8-byte words of a run; a wave covers 64 / (R / 8) runs per access when R < 512
// bytes (like the bin kernel's write-out: consecutive lanes -> consecutive bucket slots), runs of more than 512 bytes
// are walked in 512-byte pieces.  Four accesses in flight per lane.  Run starts are pseudo-random multiples of `align` inside a
// buffer of `gb` GiB (far beyond L2 + MALL).
// usage: run_bench [GiB=16] [align=8]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
enum { READ = 0, WRITE = 1, RMW = 2 };
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_runs(uint64_t *buf, uint64_t words, uint32_t run_words, uint32_t align_words,
                                              int steps, uint64_t *out) {
    const uint32_t lpr = run_words < 64 ? run_words : 64;  // lanes per run
    const int lane = (threadIdx.x & 63) % lpr;
    const uint64_t wave = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / lpr) + (threadIdx.x & 63) / lpr; // run stream id
    const uint64_t slots = (words - run_words) / align_words;
    uint64_t acc = 0;
    for (int s = 0; s < steps; s += 4) { // four runs in flight per lane group
        uint64_t base[4];
#pragma unroll
        for (int u = 0; u < 4; u++) base[u] = (mix(wave * 1000003ull + (uint64_t)(s + u)) % slots) * align_words;
        for (uint32_t o = 0; o < run_words; o += 64) {
            uint64_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = 0;
            if (o + lane < run_words && (threadIdx.x & 63) < (64 / lpr) * lpr) { // (the 64 mod lpr leftover lanes of a wave would form a partial run the host does not count)
                if (MODE != WRITE) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = buf[base[u] + o + lane];
                }
                if (MODE != READ) {
#pragma unroll
                    for (int u = 0; u < 4; u++) buf[base[u] + o + lane] = v[u] + wave;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) acc += v[u];
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
template <int MODE> double run(uint64_t *d_buf, uint64_t words, uint32_t run_bytes, uint32_t align, uint64_t *d_out) {
    const uint32_t rw = run_bytes / 8, aw = align / 8;
    const int blocks = 256 * 8 * 4;
    // about 8 GB of traffic per measurement
    const double runs_per_wave = rw < 64 ? 64.0 / rw : 1.0;
    int steps = (int)(8.0e9 / ((double)blocks * 4 * runs_per_wave * run_bytes) / (MODE == RMW ? 2 : 1));
    steps = (steps + 3) / 4 * 4;
    if (steps < 4) steps = 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_runs<MODE>, dim3(blocks), dim3(256), 0, 0, d_buf, words, rw, aw, 4, d_out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_runs<MODE>, dim3(blocks), dim3(256), 0, 0, d_buf, words, rw, aw, steps, d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 4 * runs_per_wave * steps * run_bytes * (MODE == RMW ? 2 : 1);
    hipEventDestroy(a); hipEventDestroy(b);
    return bytes / ms / 1e6; // GB/s
}
// ---- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on known byte counts (`run_bench calib`): the same run pattern with 4-
// and 8-byte words per lane, ONE dispatch per (word size, mode, run length), each touching `bytes` useful bytes exactly
// once at pseudo-random run starts.  tools/pmc_calibrate.py divides the counters of a --pmc pass by these byte counts.
template <int MODE, typename W>
__global__ void __launch_bounds__(256) k_calib(W *buf, uint64_t words, uint32_t run_words, int steps, W *out) {
    const uint32_t lpr = run_words < 64 ? run_words : 64;
    const int lane = (threadIdx.x & 63) % lpr;
    const uint64_t wave = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / lpr) + (threadIdx.x & 63) / lpr;
    const uint64_t slots = (words - run_words) / run_words; // run starts aligned to the run length
    W acc = 0;
    for (int s = 0; s < steps; s += 4) {
        uint64_t base[4];
#pragma unroll
        for (int u = 0; u < 4; u++) base[u] = (mix(wave * 1000003ull + (uint64_t)(s + u)) % slots) * run_words;
        for (uint32_t o = 0; o < run_words; o += 64) {
            W v[4] = {0, 0, 0, 0};
            if (o + lane < run_words && (threadIdx.x & 63) < (64 / lpr) * lpr) { // (leftover lanes: see k_runs)
                if (MODE != WRITE) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = buf[base[u] + o + lane];
                }
                if (MODE != READ) {
#pragma unroll
                    for (int u = 0; u < 4; u++) buf[base[u] + o + lane] = v[u] + (W)wave;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) acc += v[u];
        }
    }
    if (acc == (W)0x1234567) out[0] = acc;
}
template <int MODE, typename W> void calib(void *d_buf, uint64_t buf_bytes, uint32_t run_bytes, void *d_out, int &disp, uint32_t align_bytes = 0) {
    if (align_bytes) { // runs that start at any multiple of align_bytes: the write-out of the bin kernels (a few 8-byte messages per run)
        const uint32_t rw = run_bytes / 8, aw = align_bytes / 8;
        const int blocks = 256 * 8 * 2;
        const double runs_per_wave = rw < 64 ? 64 / rw : 1.0;
        int steps = (int)(2.0e9 / ((double)blocks * 4 * runs_per_wave * run_bytes));
        steps = (steps + 3) / 4 * 4;
        if (steps < 4) steps = 4;
        hipLaunchKernelGGL(k_runs<MODE>, dim3(blocks), dim3(256), 0, 0, (uint64_t *)d_buf, buf_bytes / 8, rw, aw, steps, (uint64_t *)d_out);
        hipDeviceSynchronize();
        const double bytes = (double)blocks * 4 * runs_per_wave * steps * run_bytes;
        printf("CALIB %d %s word=8 run=%u bytes=%.0f align=%u\n", disp++, MODE == READ ? "read" : MODE == WRITE ? "write" : "rmw", run_bytes, bytes, align_bytes);
        return;
    }
    const uint32_t rw = run_bytes / sizeof(W);
    const int blocks = 256 * 8 * 2;
    const double runs_per_wave = rw < 64 ? 64.0 / rw : 1.0;
    int steps = (int)(2.0e9 / ((double)blocks * 4 * runs_per_wave * run_bytes)); // about 2 GB per dispatch
    steps = (steps + 3) / 4 * 4;
    if (steps < 4) steps = 4;
    hipLaunchKernelGGL((k_calib<MODE, W>), dim3(blocks), dim3(256), 0, 0, (W *)d_buf, buf_bytes / sizeof(W), rw, steps, (W *)d_out);
    hipDeviceSynchronize();
    const double bytes = (double)blocks * 4 * runs_per_wave * steps * run_bytes;
    printf("CALIB %d %s word=%zu run=%u bytes=%.0f\n", disp++, MODE == READ ? "read" : MODE == WRITE ? "write" : "rmw", sizeof(W), run_bytes, bytes);
}
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "calib")) {
        const uint64_t bytes = 16ull << 30;
        void *d_buf, *d_out;
        if (hipMalloc(&d_buf, bytes) != hipSuccess || hipMalloc(&d_out, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
        hipMemset(d_buf, 0, bytes);
        hipDeviceSynchronize();
        int disp = 0;
        const uint32_t runs[] = {32, 64, 128, 256, 512, 4096, 65536};
        for (uint32_t r : runs) {
            calib<READ, uint32_t>(d_buf, bytes, r, d_out, disp);
            calib<READ, uint64_t>(d_buf, bytes, r, d_out, disp);
            calib<WRITE, uint32_t>(d_buf, bytes, r, d_out, disp);
            calib<WRITE, uint64_t>(d_buf, bytes, r, d_out, disp);
            calib<RMW, uint64_t>(d_buf, bytes, r, d_out, disp);
        }
        for (uint32_t r : {8u, 16u, 24u, 40u, 56u, 104u, 232u}) { // unaligned short runs
            calib<WRITE, uint64_t>(d_buf, bytes, r, d_out, disp, 8);
            calib<READ, uint64_t>(d_buf, bytes, r, d_out, disp, 8);
        }
        return 0;
    }
    const double gb = argc > 1 ? atof(argv[1]) : 16;
    const uint32_t align = argc > 2 ? (uint32_t)atoi(argv[2]) : 8;
    const uint64_t words = (uint64_t)(gb * 1073741824.0 / 8);
    uint64_t *d_buf, *d_out;
    if (hipMalloc(&d_buf, words * 8) != hipSuccess || hipMalloc(&d_out, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(d_buf, 0, words * 8);
    printf("buffer %.0f GiB, run starts aligned to %u bytes; GB/s of useful bytes (read-modify-write counts both directions)\n", gb, align);
    printf("%10s %12s %12s %12s\n", "run bytes", "read", "write", "rmw");
    const uint32_t runs[] = {8, 16, 32, 64, 128, 256, 512, 1024, 4096, 65536};
    for (uint32_t r : runs) {
        if (r < align && align > 8) continue;
        printf("%10u %12.0f %12.0f %12.0f\n", r, run<READ>(d_buf, words, r, align, d_out), run<WRITE>(d_buf, words, r, align, d_out),
               run<RMW>(d_buf, words, r, align, d_out));
    }
    return 0;
}
