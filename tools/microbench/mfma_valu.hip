// mfma_valu.hip — what does it cost to put the dequantization VALU of the quantized GEMM between the MFMAs of the SAME wave?
//
// Every design of the 128x128 / 256x128 kernels (in-wave pipeline, cross-stage pipeline, loader waves) lands at
// ~1.0 us per 128-k stage = 52-58 % matrix-pipe utilisation at the clock the chip sustains, whatever the DMA / LDS
// load per MFMA is.  What they have in common is the instruction mix of the compute waves:
//     v_mfma_f32_32x32x16_f16 ; 3-4 VALU (v_and_or_b32 [v_lshrrev] v_pk_add_f16 v_pk_fma_f16) ; v_mfma ... , two waves per SIMD,
// every instruction pinned with sched_barrier.  This probe runs exactly that shape without any memory traffic and
// reports cycles per MFMA slot, for 1 / 2 / 3 waves per SIMD and for each filler kind, so that the price of each
// ingredient (plain VALU, packed-fp16 VALU, the RAW edge VALU -> MFMA B operand, ds_read_b128 beside MFMAs) is known
// before the next schedule is written.
//
//   pipe-cycles per MFMA = wave cycles / (MFMAs per wave * waves per SIMD);  32 = the matrix pipe is never idle.
//
// hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu ; ./mfma_valu [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef _Float16 half_t;
typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef half_t half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum Kind { K_NONE = 0, K_ANDOR, K_SHIFT, K_PKADD, K_PKFMA, K_MOV, K_NOP, K_REAL, K_REAL_NODEP, K_DSREAD, K_REAL_DSREAD, K_FMA32, NKIND };
static const char *kind_name[NKIND] = {"none", "v_and_or_b32", "v_lshrrev_b32", "v_pk_add_f16", "v_pk_fma_f16", "v_mov_b32", "s_nop",
                                       "pairbits -> B operand", "pairbits (result unused by MFMA)", "ds_read_b128", "pairbits + ds_read_b128", "v_fma_f32"};

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// NF fillers of KIND after every MFMA.  K_REAL*: NF half2 "pairbits" (and_or [+ shift on odd ones], pk_add, pk_fma) per MFMA.
template <int KIND, int NF>
__global__ __launch_bounds__(768) void k_probe(unsigned long long *out, const uint32_t *in, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    // operands made opaque to the optimizer
    u32x4 xr[4]; uint32_t q[4], f[2][4], scratch[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xr[i] = *reinterpret_cast<const u32x4 *>(in + ((threadIdx.x * 4 + i) & 1023) * 4);
        q[i] = in[(threadIdx.x + 17 * i) & 4095]; f[0][i] = f[1][i] = 0x3c003c00u; scratch[i] = q[i] ^ 0x1234u;
    }
    uint32_t m4 = 0x000F000Fu, magic = 0x64006400u, off = 0xE408E408u /* -1032 */, s = 0x2E662E66u /* 0.1 */, c = 0xB800B800u /* -0.5 */;
    asm volatile("" : "+v"(m4), "+v"(magic), "+v"(off), "+v"(s), "+v"(c));
    float f32a = 1.0f, f32b = 0.999f; asm volatile("" : "+v"(f32a), "+v"(f32b));
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem + ((threadIdx.x * 16) & 32767);
    u32x4 dsr = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const u32x4 cw = {f[kk & 1][0], f[kk & 1][1], f[kk & 1][2], f[kk & 1][3]};
            const half8_t wf = __builtin_bit_cast(half8_t, cw);
#pragma unroll
            for (int bf = 0; bf < 4; bf++) {
                __builtin_amdgcn_sched_barrier(0);
                acc[bf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, xr[bf]), wf, acc[bf], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KIND == K_REAL || KIND == K_REAL_NODEP || KIND == K_REAL_DSREAD) {
#pragma unroll
                    for (int n = 0; n < NF; n++) {
                        uint32_t v = q[(bf + n) & 3], r;
                        if ((bf + n) & 1) asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(v) : "v"(q[(bf + n) & 3]));
                        asm volatile("v_and_or_b32 %0, %1, %2, %3\n\tv_pk_add_f16 %0, %0, %4\n\tv_pk_fma_f16 %0, %0, %5, %6"
                                     : "=&v"(r) : "v"(v), "v"(m4), "v"(magic), "v"(off), "v"(s), "v"(c));
                        if (KIND == K_REAL_NODEP) scratch[(bf + n) & 3] = r; else f[(kk + 1) & 1][(bf + n) & 3] = r;
                    }
                    if constexpr (KIND == K_REAL_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(dsr) : "v"(lds_addr) : "memory");
                } else {
#pragma unroll
                    for (int n = 0; n < NF; n++) {
                        uint32_t &x = scratch[n & 3];
                        if constexpr (KIND == K_ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(m4), "v"(magic));
                        else if constexpr (KIND == K_SHIFT) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
                        else if constexpr (KIND == K_PKADD) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(off));
                        else if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(s), "v"(c));
                        else if constexpr (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(m4));
                        else if constexpr (KIND == K_NOP) asm volatile("s_nop 0");
                        else if constexpr (KIND == K_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(dsr) : "v"(lds_addr) : "memory");
                        else if constexpr (KIND == K_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f32a) : "v"(f32b));
                    }
                }
            }
        }
        if constexpr (KIND == K_DSREAD || KIND == K_REAL_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dsr) : : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    // keep everything alive
    float sum = f32a;
#pragma unroll
    for (int i = 0; i < 4; i++) { sum += acc[i][0] + acc[i][7] + acc[i][15]; sum += (float)(scratch[i] & 3) + (float)(dsr[i] & 1); }
    if (lane == 0) { out[blockIdx.x * 32 + wave] = t1 - t0; out[blockIdx.x * 32 + 16 + wave] = r1 - r0; }   // shader cycles, 100 MHz ticks
    if (sum == 12345.678f) out[0] = 0;
}

struct Row { double mn, mean, mx, mhz; };
template <int KIND, int NF>
static Row run(int wps, int iters, unsigned long long *d_out, const uint32_t *d_in, int nblk) {
    const int threads = wps * 256;
    // 96 KB of dynamic LDS: one work-group per CU, so `wps` really is the number of waves on each SIMD
    CHECK(hipFuncSetAttribute((const void *)k_probe<KIND, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((k_probe<KIND, NF>), dim3(nblk), dim3(threads), 96 * 1024, 0, d_out, d_in, iters);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(nblk * 32);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    // Matrix-pipe arbitration between the waves of a SIMD is oldest-first: the older wave runs at full rate and the younger
    // one gets what is left, so the LAST wave to finish (max) is the one that shows the pipe's throughput.
    double mn = 1e30, mx = 0, sm = 0, cyc_sum = 0, ref_sum = 0; int n = 0;
    for (int b = 0; b < nblk; b++) for (int w = 0; w < wps * 4; w++) {
        const double cyc = (double)h[b * 32 + w] / (16.0 * iters * wps); mn = std::min(mn, cyc); mx = std::max(mx, cyc); sm += cyc; n++;
        cyc_sum += (double)h[b * 32 + w]; ref_sum += (double)h[b * 32 + 16 + w];
    }
    return Row{mn, sm / n, mx, ref_sum > 0 ? cyc_sum / ref_sum * 100.0 : 0.0};
}

template <int KIND, int NF>
static void line(int iters, unsigned long long *d_out, const uint32_t *d_in, int nblk) {
    printf("  %-34s x%d :", kind_name[KIND], NF);
    for (int wps = 1; wps <= 3; wps++) { const Row r = run<KIND, NF>(wps, iters, d_out, d_in, nblk); printf("   %dw/SIMD %5.1f (first wave %5.1f) @%4.0f MHz", wps, r.mx, r.mn, r.mhz); }
    printf("\n");
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int nblk = pr.multiProcessorCount;          // one work-group per CU: the whole chip busy, like the GEMM
    unsigned long long *d_out; uint32_t *d_in;
    CHECK(hipMalloc(&d_out, (size_t)nblk * 32 * 8)); CHECK(hipMalloc(&d_in, 4096 * 4 * 4));
    std::vector<uint32_t> h(4096 * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0x3c003800u ^ (uint32_t)(i * 2654435761u >> 20 & 0x03ff03ffu);
    CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("%s, %d CUs; pipe-cycles per v_mfma_f32_32x32x16_f16 of the LAST wave to finish (32 = matrix pipe never idle); shader-clock cycles (s_memtime); MHz = s_memtime / s_memrealtime\n", pr.gcnArchName, nblk);
    // The sustained matrix-core rate under the power limit: ~50 ms of nothing but MFMAs on random fp16 data, then on zeros.
    // TFLOP/s = 256 CUs x 4 SIMDs x 32768 flop / (pipe-cycles / clock).
    for (int zero = 0; zero < 2; zero++) {
        if (zero) { std::vector<uint32_t> z(h.size(), 0); CHECK(hipMemcpy(d_in, z.data(), z.size() * 4, hipMemcpyHostToDevice)); }
        const Row r = run<K_NONE, 0>(2, 100000, d_out, d_in, nblk);
        printf("  sustained, MFMA only, 2w/SIMD, %s inputs: %5.1f pipe-cycles/MFMA @%4.0f MHz -> %6.0f TFLOP/s dense fp16\n", zero ? "zero" : "random",
               r.mx, r.mhz, nblk * 4.0 * 32768.0 / (r.mx / (r.mhz * 1e6)) / 1e12);
    }
    CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // warm the clocks
    run<K_NONE, 0>(2, iters * 4, d_out, d_in, nblk);
    line<K_NONE, 0>(iters, d_out, d_in, nblk);
    line<K_NOP, 4>(iters, d_out, d_in, nblk);
    line<K_MOV, 4>(iters, d_out, d_in, nblk);
    line<K_ANDOR, 1>(iters, d_out, d_in, nblk); line<K_ANDOR, 4>(iters, d_out, d_in, nblk); line<K_ANDOR, 8>(iters, d_out, d_in, nblk);
    line<K_SHIFT, 4>(iters, d_out, d_in, nblk);
    line<K_PKADD, 1>(iters, d_out, d_in, nblk); line<K_PKADD, 4>(iters, d_out, d_in, nblk);
    line<K_PKFMA, 1>(iters, d_out, d_in, nblk); line<K_PKFMA, 4>(iters, d_out, d_in, nblk); line<K_PKFMA, 8>(iters, d_out, d_in, nblk);
    line<K_FMA32, 4>(iters, d_out, d_in, nblk);
    line<K_REAL, 1>(iters, d_out, d_in, nblk); line<K_REAL, 2>(iters, d_out, d_in, nblk);
    line<K_REAL_NODEP, 1>(iters, d_out, d_in, nblk); line<K_REAL_NODEP, 2>(iters, d_out, d_in, nblk);
    line<K_DSREAD, 1>(iters, d_out, d_in, nblk); line<K_DSREAD, 2>(iters, d_out, d_in, nblk);
    line<K_REAL_DSREAD, 1>(iters, d_out, d_in, nblk);
    return 0;
}
