// quantize_dev.h — the per-block bodies of the activation quantizers, shared by the standalone kernels
// (quantize_act.hip) and the fused quantize+GEMV decode kernel (gemv_q.hip).  Bit-identical to the CPU:
//   q8_K_superblock : quantize_row_q8_K_ref, src/ggml-quants.c:2479-2516 (one wave per 256 values)
//   q8_0_block      : AVX2 body of quantize_row_q8_0, src/ggml-cpu/ggml-cpu-quants.c:778-815 (REF=false) or
//                     quantize_row_q8_0_ref, src/ggml-quants.c:194-217 (REF=true); 8 lanes per 32 values
// Must be compiled with -ffp-contract=off (iscale*x rounds before the integer conversion).
#pragma once
#include "cdna4_common.h"

// lane holds e[0..3] = x[4*lane .. 4*lane+3] of one superblock; returns q[4] and the block scale d (wave-uniform)
__device__ __forceinline__ void q8_K_superblock(const float (&e)[4], int lane, int (&q)[4], float &d) {
    // first index with the largest |x| keeps its SIGNED value (src/ggml-quants.c:2485-2491)
    float amax = 0.f, mx = 0.f; int idx = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const float ax = fabsf(e[i]); if (ax > amax) { amax = ax; mx = e[i]; idx = lane * 4 + i; } }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float oa = __shfl_xor(amax, o, 64), om = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; }
    }
    q[0] = q[1] = q[2] = q[3] = 0; d = 0.f;
    if (amax != 0.f) {
        const float iscale = -127.f / mx;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int t = (int)__builtin_rintf(iscale * e[i]); q[i] = t < 127 ? t : 127; }   // nearest_int == RNE
        d = 1.0f / iscale;
    }
}

// lane holds 4 consecutive values of a 32-value block spread over 8 adjacent lanes; dh = the fp16-rounded scale
template <bool REF>
__device__ __forceinline__ void q8_0_block(const float (&e)[4], int (&q)[4], float &dh) {
    float amax = fmaxf(fmaxf(fabsf(e[0]), fabsf(e[1])), fmaxf(fabsf(e[2]), fabsf(e[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64)); amax = fmaxf(amax, __shfl_xor(amax, 2, 64)); amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const float d = amax / 127.f;
    float id;
    if (REF) id = d != 0.f ? 1.0f / d : 0.f; else id = amax != 0.f ? 127.f / amax : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = REF ? (int)roundf(e[i] * id) : (int)__builtin_rintf(e[i] * id);
    dh = h2f(f2h_bits(d));                                                  // the CPU stores d as fp16 and reads that back
}

__device__ __forceinline__ uint32_t pack4i8(const int (&q)[4]) {
    return (uint32_t)(q[0] & 0xFF) | ((uint32_t)(q[1] & 0xFF) << 8) | ((uint32_t)(q[2] & 0xFF) << 16) | ((uint32_t)(q[3] & 0xFF) << 24);
}

// One lane's 16-value chunk of a Q8_K superblock -> its 32 bytes of the GEMM's fp16 activation image (pair-interleaved: within
// every 4 consecutive k the order is k0,k2,k1,k3).  The 16 lanes of a superblock must be adjacent lanes of one wave and all
// execute this; c15 = the chunk's index within its superblock.  Exactly the arithmetic of k_quantize_q8_K (quantize_act.hip),
// used by the GEMM variant that quantizes its own activations (k_gemm_kq_w12, EXP bit 10).
__device__ __forceinline__ void q8_K_chunk16_image(const float (&e)[16], int c15, u32x4 &lo, u32x4 &hi) {
    float amax = 0.f, mx = 0.f; int idx = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { const float ax = fabsf(e[i]); if (ax > amax) { amax = ax; mx = e[i]; idx = c15 * 16 + i; } }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const float oa = __shfl_xor(amax, o, 64), om = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; }
    }
    half_t hv[16];
    if (amax != 0.f) {
        const float iscale = -127.f / mx, d = 1.0f / iscale;
#pragma unroll
        for (int i = 0; i < 16; i++) { const int v = (int)__builtin_rintf(iscale * e[i]); hv[i] = (half_t)(d * (float)(v < 127 ? v : 127)); }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) hv[i] = (half_t)0.f;
    }
    auto pk = [](half_t a, half_t b) __attribute__((always_inline)) { const half2_t t = {a, b}; return __builtin_bit_cast(uint32_t, t); };
    lo.x = pk(hv[0], hv[2]); lo.y = pk(hv[1], hv[3]); lo.z = pk(hv[4], hv[6]); lo.w = pk(hv[5], hv[7]);
    hi.x = pk(hv[8], hv[10]); hi.y = pk(hv[9], hv[11]); hi.z = pk(hv[12], hv[14]); hi.w = pk(hv[13], hv[15]);
}

// One Q8_K superblock by the 16 adjacent lanes of a DPP row -> its 512 bytes of the GEMM's k-panel-major, pair-interleaved fp16 image, as two 16-byte pieces per
// lane: lane l16 holds the elements 8 l16 .. + 7 (e[0..7]: piece o0, panel 2 sb) and 128 + 8 l16 .. + 7 (e[8..15]: piece o1, panel 2 sb + 1), so the sixteen lanes
// read two 512-byte runs and write two whole 256-byte panel rows.  The arithmetic — and therefore every bit of the image — is k_quantize_q8_K's
// (quantize_act.hip; quantize_row_q8_K_ref, src/ggml-quants.c:2479-2516): the selection of the largest |x| (signed value kept, first index wins) is commutative
// and associative, so it does not care which elements a lane scans as long as it scans them in rising index order.  Used by the one-launch prefill step
// (k_gemm_kq_t64<.., FQ>, gemm_kq_t64.inc): every work-group quantizes its share of the activation rows in front of the grid barrier.
__device__ __forceinline__ void q8_K_group16_image(const float (&e)[16], int l16, u32x4 &o0, u32x4 &o1) {
    float amax = 0.f, mx = 0.f; int idx = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { const float ax = fabsf(e[i]); if (ax > amax) { amax = ax; mx = e[i]; idx = 128 * (i >> 3) + 8 * l16 + (i & 7); } }
    auto take = [&](float oa, float om, int oi) __attribute__((always_inline)) { if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; } };
    take(dpp_f32<0xB1>(amax), dpp_f32<0xB1>(mx), dpp_i32<0xB1>(idx));
    take(dpp_f32<0x4E>(amax), dpp_f32<0x4E>(mx), dpp_i32<0x4E>(idx));
    take(dpp_f32<0x141>(amax), dpp_f32<0x141>(mx), dpp_i32<0x141>(idx));
    take(dpp_f32<0x140>(amax), dpp_f32<0x140>(mx), dpp_i32<0x140>(idx));
    half_t hv[16];
    if (amax != 0.f) {
        const float iscale = -127.f / mx, d = 1.0f / iscale;
#pragma unroll
        for (int i = 0; i < 16; i++) { const int v = (int)__builtin_rintf(iscale * e[i]); const int q = v < 127 ? v : 127; hv[i] = (half_t)(d * (float)q); }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) hv[i] = (half_t)0.f;
    }
    auto pk = [](half_t a, half_t b) __attribute__((always_inline)) { const half2_t t = {a, b}; return __builtin_bit_cast(uint32_t, t); };
    o0.x = pk(hv[0], hv[2]); o0.y = pk(hv[1], hv[3]); o0.z = pk(hv[4], hv[6]); o0.w = pk(hv[5], hv[7]);
    o1.x = pk(hv[8], hv[10]); o1.y = pk(hv[9], hv[11]); o1.z = pk(hv[12], hv[14]); o1.w = pk(hv[13], hv[15]);
}

