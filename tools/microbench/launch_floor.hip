// launch_floor.hip — what is the floor for "one launch that streams a 9.4 MB weight matrix from HBM once"?
// The decode GEMV (k_gemv_q_fused<Q4_K>, 4096 x 4096, B = 1) takes ~4.4 us per launch = 27 % of the 8 TB/s roof; this probe
// separates what no kernel of that size can avoid: (a) back-to-back launches of an EMPTY kernel of the same geometry
// (dispatch + completion), (b) the same geometry reading every byte of a matrix exactly once with the widest loads and
// nothing else (one HBM round trip + the stream), for several work-group counts, on 64 rotating matrices (604 MB > the
// 256 MB Infinity Cache, like bench.py) and on one cache-resident matrix.
// hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(float *sink) { if (sink == (float *)1) sink[0] = 0.f; }

// every thread issues all its 16-byte loads first (PER of them in flight per lane), then reduces
template <int PER>
__global__ __launch_bounds__(512) void k_stream(const u32x4 *__restrict__ w, size_t n16, float *sink) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    u32x4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) { const size_t idx = tid + (size_t)i * nthreads; v[i] = idx < n16 ? w[idx] : u32x4{0, 0, 0, 0}; }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) sink[tid & 255] = 1.f;
}

template <typename F> static double time_us(F launch, int n) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) launch(i);
    CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < n; i++) launch(i); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms * 1e3 / n;
}

template <int PER> static void row(const uint8_t *big, size_t mat, int ncopy, float *sink, int threads) {
    const size_t n16 = mat / 16;
    const int grid = (int)((n16 + (size_t)threads * PER - 1) / ((size_t)threads * PER));
    const double cold = time_us([&](int i) { hipLaunchKernelGGL((k_stream<PER>), dim3(grid), dim3(threads), 0, 0, (const u32x4 *)(big + (size_t)(i % ncopy) * mat), n16, sink); }, 512);
    const double warm = time_us([&](int i) { hipLaunchKernelGGL((k_stream<PER>), dim3(grid), dim3(threads), 0, 0, (const u32x4 *)big, n16, sink); }, 512);
    printf("  stream %2d x 16 B per lane, %5d work-groups of %d : cold %6.2f us = %5.2f TB/s    cache-resident %6.2f us = %5.2f TB/s\n",
           PER, grid, threads, cold, mat / cold / 1e6, warm, mat / warm / 1e6);
}

int main() {
    const size_t mat = (size_t)4096 * 4096 / 256 * 144;          // the Q4_K 4096 x 4096 matrix: 9,437,184 B
    const int ncopy = 64;
    uint8_t *big; float *sink;
    CHECK(hipMalloc(&big, mat * ncopy)); CHECK(hipMemset(big, 1, mat * ncopy)); CHECK(hipMalloc(&sink, 4096));
    printf("one launch per step, HIP events around 512 back-to-back launches; matrix = %zu B\n", mat);
    for (int g : {1, 256, 1024, 4096})
        printf("  empty kernel, %4d work-groups of 512 : %6.2f us per launch\n", g, time_us([&](int) { hipLaunchKernelGGL(k_empty, dim3(g), dim3(512), 0, 0, sink); }, 2000));
    row<1>(big, mat, ncopy, sink, 512); row<2>(big, mat, ncopy, sink, 512); row<4>(big, mat, ncopy, sink, 512); row<8>(big, mat, ncopy, sink, 512);
    row<4>(big, mat, ncopy, sink, 256); row<8>(big, mat, ncopy, sink, 256); row<16>(big, mat, ncopy, sink, 256);
    return 0;
}
