// l2_stream.hip — how fast can ONE CU pull L2/MALL-resident data (a) into LDS with global_load_lds_dwordx4 and
// (b) into VGPRs with global_load_dwordx4, as a function of waves per CU and DMA pieces in flight per wave?
// Every work-group re-reads the same `span` bytes `iters` times (span << L2), so after the first pass all hits.
// Used to size the operand-delivery budget of the MFMA GEMM.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int DEPTH>   // DEPTH x 1 KiB pieces in flight per wave
__global__ __launch_bounds__(1024) void k_dma(const char *src, size_t span, int iters, int shared_span, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char *base = src + (shared_span ? 0 : (size_t)blockIdx.x * span);
    char *lbase = smem + wave * DEPTH * 1024;
    const size_t per_iter = (size_t)nw * DEPTH * 1024;
    for (int it = 0; it < iters; it++) {
        size_t off = ((size_t)it * per_iter) % span;
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + off + (size_t)(wave * DEPTH + d) * 1024 + lane * 16), (lds_void_t *)(lbase + d * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = (float)smem[7];
}

// same as k_dma but with the scalar-base + 32-bit lane offset addressing form (saddr), M0 set by hand
template <int DEPTH>
__global__ __launch_bounds__(1024) void k_dma_saddr(const char *src, size_t span, int iters, int shared_span, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const char *base = src + (shared_span ? 0 : (size_t)blockIdx.x * span);
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem + wave * DEPTH * 1024;
    const size_t per_iter = (size_t)nw * DEPTH * 1024;
    const unsigned voff = lane * 16;
    for (int it = 0; it < iters; it++) {
        size_t off = ((size_t)it * per_iter) % span;
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const char *sb = base + off + (size_t)(wave * DEPTH + d) * 1024;
            const unsigned m0v = lbase + d * 1024;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(m0v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = (float)smem[7];
}

template <int DEPTH>
__global__ __launch_bounds__(1024) void k_reg(const char *src, size_t span, int iters, int shared_span, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char *base = src + (shared_span ? 0 : (size_t)blockIdx.x * span);
    const size_t per_iter = (size_t)nw * DEPTH * 1024;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        size_t off = ((size_t)it * per_iter) % span;
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) v[d] = *(const uint4 *)(base + off + (size_t)(wave * DEPTH + d) * 1024 + lane * 16);
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc ^= v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = 1.f;
}

template <typename F> static float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const int ncu = 256;
    const size_t span = 64 << 10;                 // 64 KiB per work-group: 16 MiB total footprint, L2/MALL resident
    char *src; float *sink; hipMalloc(&src, (size_t)ncu * 8 * span + (1 << 20)); hipMalloc(&sink, 4096 * 4);
    hipMemset(src, 1, (size_t)ncu * 8 * span);
    printf("kind,waves_per_cu,depth,shared,GBps_per_cu,TBps_chip\n");
    for (int shared = 0; shared < 2; shared++)
    for (int nw : {1, 2, 4, 8, 16}) {
        const int iters = 2048 / nw;
#define RUN(KIND, K, D) { const double bytes = (double)ncu * iters * nw * D * 1024; \
            float ms = timeit([&] { hipLaunchKernelGGL(K<D>, dim3(ncu), dim3(nw * 64), nw * D * 1024, 0, src, span, iters, shared, sink); }, 5); \
            printf("%s,%d,%d,%d,%.1f,%.2f\n", KIND, nw, D, shared, bytes / ncu / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12); }
        RUN("lds_dma", k_dma, 1) RUN("lds_dma", k_dma, 4) RUN("lds_dma", k_dma, 8)
        RUN("lds_dma_saddr", k_dma_saddr, 1) RUN("lds_dma_saddr", k_dma_saddr, 4) RUN("lds_dma_saddr", k_dma_saddr, 8)
        RUN("vgpr", k_reg, 1) RUN("vgpr", k_reg, 4) RUN("vgpr", k_reg, 8)
    }
    return 0;
}
