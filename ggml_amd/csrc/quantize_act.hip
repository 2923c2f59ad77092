// quantize_act.hip — activation quantizers, bit-identical to what the reference CPU backend does to
// src1 before its integer dot products (ggml_compute_forward_mul_mat, src/ggml-cpu/ggml-cpu.c:7490-7509):
//   K-quant weights  -> Q8_K  (quantize_row_q8_K_ref, src/ggml-quants.c:2479-2516)
//   Q4_0/Q8_0 weights-> Q8_0  (the AVX2 body of quantize_row_q8_0, src/ggml-cpu/ggml-cpu-quants.c:778-815;
//                              `ref` = quantize_row_q8_0_ref, src/ggml-quants.c:194-217, for CPY f32->q8_0)
// Output is a structure-of-arrays workspace (ours, never exposed): int8 qs[B][K], float d[B][K/QK],
// int16 bsums[B][K/16] (Q8_K only) for the int8-dot GEMV path, and/or the dequantized value d*q rounded
// to fp16 for the MFMA GEMM path.  The fp16 image has two layout twists, both private to this library:
//  * PAIR-INTERLEAVED: within every 4 consecutive k the order is (k0,k2,k1,k3) — the order the nibble/byte
//    unpackers of gemm_q_mfma.hip produce;
//  * K-PANEL-MAJOR: element (b, k) lives at ((k/128)*B + b)*128 + k%128, i.e. [K/128 panels][B rows][128 halves].
//    A row-major [B][K] image has an 8 KiB row stride at K=4096: every tile row of a GEMM stage then falls on the
//    same L2 channel and all 256 CUs walk K in lockstep — measured 19 GB/s per CU of LDS-DMA against 108 GB/s
//    for contiguous 32 KiB stage blocks (tools/microbench/l2_stream.hip).  Panel-major makes a stage contiguous.
// Compiled with -ffp-contract=off: iscale*x must round before the integer conversion, as on the CPU.
#include "cdna4_common.h"
#include "cdna4_kernels.h"

__device__ __forceinline__ u32x2 pack4h(half_t a, half_t b, half_t c, half_t d) {
    const half2_t lo = {a, b}, hi = {c, d};
    u32x2 r; r.x = __builtin_bit_cast(uint32_t, lo); r.y = __builtin_bit_cast(uint32_t, hi); return r;
}

// one wave per 256-element superblock, 4 consecutive elements per lane
__global__ __launch_bounds__(256) void k_quantize_q8_K(const float *__restrict__ x, int64_t x_row_stride, int K, int B,
                                                       int8_t *__restrict__ qs, float *__restrict__ dd,
                                                       int16_t *__restrict__ bsums, half_t *__restrict__ xh) {
    const int lane = threadIdx.x & 63;
    const int nsb = K / QK_K;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // superblock id over [B][nsb]
    if (blk >= (int64_t)B * nsb) return;
    const int b = (int)(blk / nsb), sb = (int)(blk % nsb);
    const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)b * x_row_stride + (int64_t)sb * QK_K + lane * 4);
    const float e[4] = {v.x, v.y, v.z, v.w};
    // first index with the largest |x| keeps its SIGNED value (src/ggml-quants.c:2485-2491)
    float amax = 0.f, mx = 0.f; int idx = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const float ax = fabsf(e[i]); if (ax > amax) { amax = ax; mx = e[i]; idx = lane * 4 + i; } }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float oa = __shfl_xor(amax, o, 64), om = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; }
    }
    int q[4] = {0, 0, 0, 0}; float d = 0.f;
    if (amax != 0.f) {
        const float iscale = -127.f / mx;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int t = (int)__builtin_rintf(iscale * e[i]); q[i] = t < 127 ? t : 127; }   // nearest_int == RNE
        d = 1.0f / iscale;
    }
    const int64_t base = (int64_t)b * K + (int64_t)sb * QK_K + lane * 4;
    if (qs) {
        const uint32_t packed = (uint32_t)(q[0] & 0xFF) | ((uint32_t)(q[1] & 0xFF) << 8) | ((uint32_t)(q[2] & 0xFF) << 16) | ((uint32_t)(q[3] & 0xFF) << 24);
        *reinterpret_cast<uint32_t *>(qs + base) = packed;
        if (lane == 0) dd[(int64_t)b * nsb + sb] = d;
        int s = q[0] + q[1] + q[2] + q[3];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        if ((lane & 3) == 0) bsums[(int64_t)b * (K / 16) + sb * 16 + (lane >> 2)] = (int16_t)s;
    }
    if (xh) {
        half_t h[4];
#pragma unroll
        for (int i = 0; i < 4; i++) h[i] = (half_t)(d * (float)q[i]);
        const int64_t k = (int64_t)sb * QK_K + lane * 4;
        *reinterpret_cast<u32x2 *>(xh + ((k >> 7) * B + b) * 128 + (k & 127)) = pack4h(h[0], h[2], h[1], h[3]);   // pair-interleaved, panel-major
    }
}

// 8 lanes per 32-element block, 4 consecutive elements per lane.  REF=false: AVX2 semantics
// (d = amax/127 -> fp16, id = 127/amax, RNE); REF=true: _ref semantics (id = 1/d, roundf ties away).
template <bool REF>
__global__ __launch_bounds__(256) void k_quantize_q8_0(const float *__restrict__ x, int64_t x_row_stride, int K, int B,
                                                       int8_t *__restrict__ qs, float *__restrict__ dd, half_t *__restrict__ xh) {
    const int nb = K / 32;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;             // one thread per 4 elements
    const int64_t blk = t >> 3;
    if (blk >= (int64_t)B * nb) return;                                     // whole 8-lane groups drop out together
    const int b = (int)(blk / nb), ib = (int)(blk % nb), sub = (int)(t & 7);
    const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)b * x_row_stride + (int64_t)ib * 32 + sub * 4);
    const float e[4] = {v.x, v.y, v.z, v.w};
    float amax = fmaxf(fmaxf(fabsf(e[0]), fabsf(e[1])), fmaxf(fabsf(e[2]), fabsf(e[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64)); amax = fmaxf(amax, __shfl_xor(amax, 2, 64)); amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const float d = amax / 127.f;
    float id;
    if (REF) id = d != 0.f ? 1.0f / d : 0.f; else id = amax != 0.f ? 127.f / amax : 0.f;
    int q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = REF ? (int)roundf(e[i] * id) : (int)__builtin_rintf(e[i] * id);
    const float dh = h2f(f2h_bits(d));                                      // the CPU stores d as fp16 and reads that back
    const int64_t base = (int64_t)b * K + (int64_t)ib * 32 + sub * 4;
    if (qs) {
        const uint32_t packed = (uint32_t)(q[0] & 0xFF) | ((uint32_t)(q[1] & 0xFF) << 8) | ((uint32_t)(q[2] & 0xFF) << 16) | ((uint32_t)(q[3] & 0xFF) << 24);
        *reinterpret_cast<uint32_t *>(qs + base) = packed;
        if (sub == 0) dd[(int64_t)b * nb + ib] = dh;
    }
    if (xh) {
        half_t h[4];
#pragma unroll
        for (int i = 0; i < 4; i++) h[i] = (half_t)(dh * (float)q[i]);
        const int64_t k = (int64_t)ib * 32 + sub * 4;
        *reinterpret_cast<u32x2 *>(xh + ((k >> 7) * B + b) * 128 + (k & 127)) = pack4h(h[0], h[2], h[1], h[3]);
    }
}

int cdna4_launch_quantize_q8_K(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d,
                               int16_t *bsums, void *xh, hipStream_t st) {
    if (K % QK_K) return cdna4_set_error_msg("quantize_q8_K: K must be a multiple of 256");
    if (B == 0 || K == 0) return 0;
    const int64_t nblk = B * (K / QK_K);
    hipLaunchKernelGGL(k_quantize_q8_K, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, st, x, x_row_stride, (int)K, (int)B, qs, d, bsums, (half_t *)xh);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int cdna4_launch_quantize_q8_0(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d,
                               void *xh, bool ref_rounding, hipStream_t st) {
    if (K % 32) return cdna4_set_error_msg("quantize_q8_0: K must be a multiple of 32");
    if (B == 0 || K == 0) return 0;
    const int64_t nthr = B * (K / 4);
    const dim3 grid((unsigned)((nthr + 255) / 256));
    if (ref_rounding) hipLaunchKernelGGL(k_quantize_q8_0<true>, grid, dim3(256), 0, st, x, x_row_stride, (int)K, (int)B, qs, d, (half_t *)xh);
    else hipLaunchKernelGGL(k_quantize_q8_0<false>, grid, dim3(256), 0, st, x, x_row_stride, (int)K, (int)B, qs, d, (half_t *)xh);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
