// exact.hip — GGML_CDNA4_EXACT: the three ops of a gpt-2 graph whose fp32 SUMMATION ORDER differs from the reference CPU backend's by default
// (MUL_MAT 2e-7, NORM 1e-7, SOFT_MAX 1e-7 per op: profiles/r03/gpt2_parity.jsonl) re-done in the CPU's own order, so that the whole chain —
// every other op on the path is already bit-identical — reproduces the CPU backend's logits bit for bit (BASELINE.json configs[3]: "logits vs CPU
// <= 1e-3"; the reference amplifies a 1e-6 perturbation of one LayerNorm gain to 1.7e-2, so nothing short of identical bits can meet it).
// "The CPU's order" means the x86-64-v3 build of oracle/ref.mk (AVX2 + FMA, gcc 11 -O3), function by function:
//   ggml_vec_dot_q4_0_q8_0 / _q8_0_q8_0   src/ggml-cpu/ggml-cpu-quants.c:2005-2028, 3520-3536: eight fp32 lane accumulators (lane L = elements 4L .. 4L+3 of
//                                          every 32-block), acc = fma(d_w * d_x, (float)isum, acc) block after block, then hsum_float_8 (:49-55)
//   ggml_vec_dot_f32                       src/ggml-cpu/ggml-cpu.c:1346-1377: four 8-lane accumulators over 32-element steps (fma), GGML_F32x8_REDUCE (:670-688),
//                                          and the leftover loop AS GCC COMPILED IT (objdump of oracle/_ref/libggml-cpu.so): groups of 8 and 4 multiply then add
//                                          in index order, the last <= 3 elements are scalar FMAs
//   ggml_compute_forward_norm_f32          ggml-cpu.c:6929-6978: sequential double sums
//   ggml_compute_forward_soft_max_f32      ggml-cpu.c:8848-8944 with ggml_vec_soft_max_f32 (:2041-2092): ggml_v_expf (:1912-1949) on chunks of 8, each chunk's
//                                          sum in the order of the AVX2 shuffles, accumulated in double; the row's tail through glibc 2.35's expf
//                                          (sysdeps/ieee754/flt-32/e_expf.c, restated below and checked against libm bit for bit by tests/test_exact_math.py)
// Opt-in, for verification: these kernels trade speed for order (one lane per row for NORM / SOFT_MAX, serial block loops for MUL_MAT).
#include "../../include/ggml_cdna4.h"
#include "cdna4_common.h"
#include "cdna4_kernels.h"
#include "exact_math.h"
#include <math.h>

typedef ggml_cdna4_tensor T4;
#define NEEDX(c, msg) do { if (!(c)) return cdna4_set_error_msg("exact: " msg); } while (0)

// ------------------------------------------------------------------------------------------------ MUL_MAT, Q4_0 / Q8_0 weights x Q8_0 activations
// wave = one weight row x eight activation columns; lane = (column c = lane >> 3, accumulator L = lane & 7)
template <int TYPE>
__global__ __launch_bounds__(256) void k_mul_mat_exact_q(const uint8_t *__restrict__ W, int64_t w_row_bytes, const int8_t *__restrict__ qs, const float *__restrict__ dd,
                                                        float *__restrict__ Y, int64_t y_row, int M, int K, int B) {
    const int lane = threadIdx.x & 63, L = lane & 7, c = lane >> 3;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ngrp = (B + 7) / 8;
    if (unit >= (int64_t)M * ngrp) return;
    const int m = (int)(unit / ngrp), b = (int)(unit % ngrp) * 8 + c;
    const int nb = K / 32;
    const int bb = b < B ? b : B - 1;                                   // (lanes past B compute a duplicate: the shuffles below need all 64)
    const uint8_t *wrow = W + (int64_t)m * w_row_bytes;
    const int8_t *xq = qs + (int64_t)bb * K;
    const float *xd = dd + (int64_t)bb * nb;
    float acc = 0.f;
    for (int ib = 0; ib < nb; ib++) {
        const uint8_t *blk = wrow + (int64_t)ib * QT<TYPE>::BYTES;
        const float dw = h2f(ld_u16(blk));
        const int yq = *reinterpret_cast<const int *>(xq + ib * 32 + 4 * L);
        int isum;
        if constexpr (TYPE == CDNA4_Q4_0) {
            // element e < 16 = low nibble of qs[e], e >= 16 = high nibble of qs[e - 16] (bytes_from_nibbles_32): lane L takes bytes 4 (L & 3) .. + 3
            const uint32_t q = ld_u32_a2(blk + 2 + 4 * (L & 3));
            const int v = (int)((L < 4 ? q : q >> 4) & 0x0F0F0F0Fu);
            isum = __builtin_amdgcn_sdot4(v, yq, 0, false) - 8 * __builtin_amdgcn_sdot4(0x01010101, yq, 0, false);
        } else {
            const int v = (int)ld_u32_a2(blk + 2 + 4 * L);
            isum = __builtin_amdgcn_sdot4(v, yq, 0, false);
        }
        const float d = dw * xd[ib];                                    // GGML_FP16_TO_FP32(x.d) * GGML_FP16_TO_FP32(y.d): an fp32 product
        acc = __builtin_fmaf(d, (float)isum, acc);                      // _mm256_fmadd_ps(d, q, acc)
    }
    // hsum_float_8: (a[i] + a[i + 4]), then (.. [0] + [2], [1] + [3]), then [0] + [1]
    acc = acc + __shfl_xor(acc, 4, 64);
    acc = acc + __shfl_xor(acc, 2, 64);
    acc = acc + __shfl_xor(acc, 1, 64);
    if (L == 0 && b < B) Y[(int64_t)b * y_row + m] = acc;
}

// ------------------------------------------------------------------------------------------------ MUL_MAT, Q4_K / Q5_K / Q6_K weights x Q8_K activations
// The AVX2 bodies of ggml_vec_dot_q4_K_q8_K / _q5_K_q8_K / _q6_K_q8_K (src/ggml-cpu/ggml-cpu-quants.c:5712-5775, 6283-6364, 6941-7018).  Same wave shape as above:
// lane L is 32-bit lane L of the __m256i `sumi` — bytes 4L .. 4L+3 of every 32-byte chunk, scaled by that chunk's 6-bit (Q4_K / Q5_K) or int8 (Q6_K: one scale per
// 16 elements, so lanes 0-3 and 4-7 differ) scale; the integer sums are exact in any order (maddubs cannot saturate: <= 2*63*127), then per superblock
// acc = fma(y.d * fp16(x.d), (float)sumi, acc) and hsum_float_8 at the end.  The mins: Q4_K keeps FOUR fp32 accumulators acc_m[k] = fma(dmin, (float)(mins[2k] * s[2k] +
// mins[2k+1] * s[2k+1]), acc_m[k]) with s[j] = bsums[2j] + bsums[2j+1] (hadd_epi16 / madd_epi16), folded [0]+[2], [1]+[3], [0]+[1] and added LAST; Q5_K keeps ONE
// scalar, summs = fma(dmin, (float)(sum of the four), summs) — gcc contracts `summs += dmin * x` (vfmadd231ss in oracle/_ref/libggml-cpu.so).
template <int TYPE>
__global__ __launch_bounds__(256) void k_mul_mat_exact_kq(const uint8_t *__restrict__ W, int64_t w_row_bytes, const int8_t *__restrict__ qs, const float *__restrict__ dd,
                                                         const int16_t *__restrict__ bsums, float *__restrict__ Y, int64_t y_row, int M, int K, int B) {
    const int lane = threadIdx.x & 63, L = lane & 7, c = lane >> 3;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ngrp = (B + 7) / 8;
    if (unit >= (int64_t)M * ngrp) return;
    const int m = (int)(unit / ngrp), b = (int)(unit % ngrp) * 8 + c;
    const int nsb = K / QK_K;
    const int bb = b < B ? b : B - 1;
    const uint8_t *wrow = W + (int64_t)m * w_row_bytes;
    const int8_t *xq = qs + (int64_t)bb * K;
    const float *xd = dd + (int64_t)bb * nsb;
    const int16_t *xs = bsums + (int64_t)bb * (K / 16);
    constexpr uint32_t M4 = 0x0F0F0F0Fu;
    float acc = 0.f, am = 0.f;
    for (int i = 0; i < nsb; i++) {
        const uint8_t *blk = wrow + (int64_t)i * QT<TYPE>::BYTES;
        const int8_t *q8 = xq + i * QK_K + 4 * L;
        const float yd = xd[i];
        int sumi = 0;
        float d;
        if constexpr (TYPE == CDNA4_Q6_K) {                             // block_q6_K: ql[128] qh[64] scales[16] d
            d = yd * h2f(ld_u16(blk + 208));
            const int8_t *sc = reinterpret_cast<const int8_t *>(blk + 192) + (L >> 2);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t l1 = ld_u32_a2(blk + 64 * j + 4 * L), l2 = ld_u32_a2(blk + 64 * j + 32 + 4 * L), h = ld_u32_a2(blk + 128 + 32 * j + 4 * L);
                const uint32_t q[4] = {(l1 & M4) | ((h & 0x03030303u) << 4), (l2 & M4) | (((h >> 2) & 0x03030303u) << 4),
                                       ((l1 >> 4) & M4) | (((h >> 4) & 0x03030303u) << 4), ((l2 >> 4) & M4) | (((h >> 6) & 0x03030303u) << 4)};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int y = *reinterpret_cast<const int *>(q8 + 128 * j + 32 * k);
                    const int p = __builtin_amdgcn_sdot4((int)q[k], y, 0, false) - __builtin_amdgcn_sdot4(0x20202020, y, 0, false);
                    sumi += (int)sc[8 * j + 2 * k] * p;
                }
            }
        } else {                                                        // block_q4_K: d dmin scales[12] qs[128]; block_q5_K: d dmin scales[12] qh[32] qs[128]
            d = yd * h2f(ld_u16(blk));
            const float dmin = -yd * h2f(ld_u16(blk + 2));
            uint32_t u0 = ld_u32_a2(blk + 4), u1 = ld_u32_a2(blk + 8), u2 = ld_u32_a2(blk + 12);
            const uint32_t m1 = ((u2 >> 4) & M4) | (((u1 >> 6) & 0x03030303u) << 4);          // utmp[3]: mins 4 .. 7
            const uint32_t m0 = u1 & 0x3F3F3F3Fu;                                              // utmp[2]: mins 0 .. 3
            const uint32_t s1 = (u2 & M4) | (((u0 >> 6) & 0x03030303u) << 4);                 // utmp[1]: scales 4 .. 7
            const uint32_t s0 = u0 & 0x3F3F3F3Fu;                                              // utmp[0]: scales 0 .. 3
            const uint64_t sc = ((uint64_t)s1 << 32) | s0, mn = ((uint64_t)m1 << 32) | m0;
            const uint8_t *qp = blk + (TYPE == CDNA4_Q5_K ? 48 : 16) + 4 * L;
            uint32_t hb = 0;
            if constexpr (TYPE == CDNA4_Q5_K) hb = ld_u32_a2(blk + 16 + 4 * L);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t q = ld_u32_a2(qp + 32 * j);
                uint32_t lo = q & M4, hi = (q >> 4) & M4;
                if constexpr (TYPE == CDNA4_Q5_K) { lo |= ((hb >> (2 * j)) & 0x01010101u) << 4; hi |= ((hb >> (2 * j + 1)) & 0x01010101u) << 4; }
                const int yl = *reinterpret_cast<const int *>(q8 + 64 * j), yh = *reinterpret_cast<const int *>(q8 + 64 * j + 32);
                sumi += (int)((sc >> (16 * j)) & 0xFF) * __builtin_amdgcn_sdot4((int)lo, yl, 0, false) + (int)((sc >> (16 * j + 8)) & 0xFF) * __builtin_amdgcn_sdot4((int)hi, yh, 0, false);
            }
            // prod[k] = mins[2k] * (bsums[4k] + bsums[4k+1]) + mins[2k+1] * (bsums[4k+2] + bsums[4k+3])
            int prod[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int16_t *s = xs + i * 16 + 4 * k;
                prod[k] = (int)((mn >> (16 * k)) & 0xFF) * ((int)s[0] + (int)s[1]) + (int)((mn >> (16 * k + 8)) & 0xFF) * ((int)s[2] + (int)s[3]);
            }
            if constexpr (TYPE == CDNA4_Q4_K) {
                const int k = L & 3;
                const int pk = k == 0 ? prod[0] : k == 1 ? prod[1] : k == 2 ? prod[2] : prod[3];
                am = __builtin_fmaf(dmin, (float)pk, am);               // _mm_fmadd_ps(set1(dmin), cvtepi32_ps(prod), acc_m): lane k of acc_m lives in lanes L = k, k + 4
            } else {
                am = __builtin_fmaf(dmin, (float)(prod[0] + prod[1] + prod[2] + prod[3]), am);       // summs += dmin * hsum
            }
        }
        acc = __builtin_fmaf(d, (float)sumi, acc);
    }
    acc = acc + __shfl_xor(acc, 4, 64);                                 // hsum_float_8
    acc = acc + __shfl_xor(acc, 2, 64);
    acc = acc + __shfl_xor(acc, 1, 64);
    if constexpr (TYPE == CDNA4_Q4_K) {
        am = am + __shfl_xor(am, 2, 64);                                // _mm_add_ps(acc_m, _mm_movehl_ps(acc_m, acc_m))
        am = am + __shfl_xor(am, 1, 64);                                // _mm_add_ss(acc_m, _mm_movehdup_ps(acc_m))
    }
    if constexpr (TYPE != CDNA4_Q6_K) acc = acc + am;
    if (L == 0 && b < B) Y[(int64_t)b * y_row + m] = acc;
}

// ------------------------------------------------------------------------------------------------ MUL_MAT, F32 x F32
// half a wave (32 lanes = the four 8-lane accumulators) per output element
__global__ __launch_bounds__(256) void k_mul_mat_exact_f32(const T4 a, const T4 b, const T4 d, int64_t nout) {
    const int64_t o = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    const int64_t oo = o < nout ? o : nout - 1;
    int64_t i = oo;
    const int64_t i0 = i % d.ne[0]; i /= d.ne[0]; const int64_t i1 = i % d.ne[1]; i /= d.ne[1]; const int64_t i2 = i % d.ne[2], i3 = i / d.ne[2];
    const int64_t r2 = b.ne[2] / a.ne[2], r3 = b.ne[3] / a.ne[3];
    const float *x = (const float *)((const char *)a.data + i0 * a.nb[1] + (i2 / r2) * a.nb[2] + (i3 / r3) * a.nb[3]);
    const float *y = (const float *)((const char *)b.data + i1 * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3]);
    const int n = (int)a.ne[0], np = n & ~31;
    float acc = 0.f;
    for (int k = 0; k < np; k += 32) acc = __builtin_fmaf(x[k + l32], y[k + l32], acc);       // sum[j] lane l: element k + 8 j + l = k + l32
    // GGML_F32x8_REDUCE: x[0] += x[2], x[1] += x[3]; x[0] += x[1]; lanes i + (i + 4); hadd; hadd
    acc = acc + __shfl_xor(acc, 16, 64);
    acc = acc + __shfl_xor(acc, 8, 64);
    acc = acc + __shfl_xor(acc, 4, 64);
    acc = acc + __shfl_xor(acc, 1, 64);
    acc = acc + __shfl_xor(acc, 2, 64);
    if (l32 == 0 && o < nout) {
        float sumf = acc;
        int k = np, r = n - np;
        while (r >= 8) {                                                // vmulps ymm, then eight vaddss in index order
            float pr[8];
            for (int e = 0; e < 8; e++) pr[e] = x[k + e] * y[k + e];
            for (int e = 0; e < 8; e++) sumf = sumf + pr[e];
            k += 8; r -= 8;
        }
        if (r >= 4) {                                                   // vmulps xmm, four vaddss
            float pr[4];
            for (int e = 0; e < 4; e++) pr[e] = x[k + e] * y[k + e];
            for (int e = 0; e < 4; e++) sumf = sumf + pr[e];
            k += 4; r -= 4;
        }
        for (; r > 0; r--, k++) sumf = __builtin_fmaf(x[k], y[k], sumf);       // vfmadd231ss
        *(float *)((char *)d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = sumf;
    }
}

// ------------------------------------------------------------------------------------------------ NORM: one lane per row, sequential double sums
__global__ __launch_bounds__(64) void k_norm_exact(const T4 a, const T4 d, float eps, int64_t nrow) {
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= nrow) return;
    const int64_t i1 = row % a.ne[1], i2 = (row / a.ne[1]) % a.ne[2], i3 = row / (a.ne[1] * a.ne[2]);
    const float *x = (const float *)((const char *)a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float *y = (float *)((char *)d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int64_t n = a.ne[0];
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) sum += (double)x[i];
    const float mean = (float)(sum / (double)n);
    double sum2 = 0.0;
    for (int64_t i = 0; i < n; i++) { const float v = x[i] - mean; y[i] = v; sum2 += (double)(v * v); }
    const float variance = (float)(sum2 / (double)n);
    const float scale = 1.0f / sqrtf(variance + eps);
    for (int64_t i = 0; i < n; i++) y[i] = y[i] * scale;               // ggml_vec_scale_f32
}

// ------------------------------------------------------------------------------------------------ RMS_NORM: one lane per row (ggml_compute_forward_rms_norm_f32, ggml-cpu.c:7000-7046)
__global__ __launch_bounds__(64) void k_rms_norm_exact(const T4 a, const T4 d, float eps, int64_t nrow) {
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= nrow) return;
    const int64_t i1 = row % a.ne[1], i2 = (row / a.ne[1]) % a.ne[2], i3 = row / (a.ne[1] * a.ne[2]);
    const float *x = (const float *)((const char *)a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float *y = (float *)((char *)d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int64_t n = a.ne[0];
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) { const float sq = x[i] * x[i]; sum += (double)sq; }       // sum += (ggml_float)(x * x): an fp32 product, a double sum
    const float mean = (float)(sum / (double)n);
    const float scale = 1.0f / sqrtf(mean + eps);
    for (int64_t i = 0; i < n; i++) y[i] = x[i] * scale;               // memcpy + ggml_vec_scale_f32
}

// ------------------------------------------------------------------------------------------------ SILU: ggml_vec_silu_f32 (ggml-cpu.c:2017-2039) — per ROW, chunks of eight through
// ggml_v_silu (:1952-1959: x / (1 + ggml_v_expf(0 - x))), the row's last n % 8 elements through x / (1.0f + expf(-x)) with glibc's expf
__global__ __launch_bounds__(256) void k_silu_exact(const float *__restrict__ x, float *__restrict__ y, int nc, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % nc);
    const float v = x[i];
    const float e = col < (nc & ~7) ? exact_v_expf(0.0f - v) : exact_expf_glibc(-v);
    y[i] = v / (1.0f + e);
}

// ------------------------------------------------------------------------------------------------ SOFT_MAX (no mask, no ALiBi): one lane per row
__global__ __launch_bounds__(64) void k_soft_max_exact(const float *__restrict__ x, float *__restrict__ y, int nc, int64_t nrow, float scale) {
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= nrow) return;
    const float *sp = x + row * nc;
    float *dp = y + row * nc;
    float mx = -INFINITY;
    for (int i = 0; i < nc; i++) { const float v = sp[i] * scale; dp[i] = v; mx = fmaxf(mx, v); }      // wp = sp * scale (dp as the scratch row); ggml_vec_max_f32
    double sum = 0.0;
    int i = 0;
    for (; i + 7 < nc; i += 8) {
        float val[8];
        for (int l = 0; l < 8; l++) { val[l] = exact_v_expf(dp[i + l] - mx); dp[i + l] = val[l]; }
        float v2[4];
        for (int l = 0; l < 4; l++) v2[l] = val[4 + l] + val[l];        // _mm_add_ps(extractf128(val, 1), castps256_ps128(val))
        v2[0] = v2[0] + v2[2]; v2[1] = v2[1] + v2[3];                   // _mm_add_ps(val2, _mm_movehl_ps(val2, val2))
        sum += (double)(v2[0] + v2[1]);                                 // _mm_add_ss(val2, _mm_movehdup_ps(val2))
    }
    for (; i < nc; i++) { const float v = exact_expf_glibc(dp[i] - mx); sum += (double)v; dp[i] = v; }
    const float inv = (float)(1.0 / sum);
    for (i = 0; i < nc; i++) dp[i] = dp[i] * inv;                       // ggml_vec_scale_f32(nc, dp, sum)
}

extern "C" {

static bool exact_kq(int type) { return type == CDNA4_Q4_K || type == CDNA4_Q5_K || type == CDNA4_Q6_K; }
static bool exact_q0(int type) { return type == CDNA4_Q4_0 || type == CDNA4_Q8_0; }
// workspace: int8 qs [B][K] | float d [B][K / block] | (K-quants) int16 bsums [B][K / 16]
static size_t exact_ws_d(int64_t K, int64_t B) { return (size_t)((B * K + 255) & ~(int64_t)255); }
static size_t exact_ws_bsums(int type, int64_t K, int64_t B) { return exact_ws_d(K, B) + (size_t)((B * (K / (exact_kq(type) ? QK_K : 32)) * 4 + 255) & ~(int64_t)255); }
int ggml_cdna4_mul_mat_exact_supported(int type, int64_t K) { return K > 0 && ((exact_q0(type) && K % 32 == 0) || (exact_kq(type) && K % QK_K == 0)); }
size_t ggml_cdna4_mul_mat_exact_workspace_size(int type, int64_t K, int64_t B) {
    if (!ggml_cdna4_mul_mat_exact_supported(type, K) || B <= 0) return 0;
    return exact_ws_bsums(type, K, B) + (exact_kq(type) ? (size_t)B * (K / 16) * 2 : 0) + 256;
}
// ggml_compute_forward_mul_mat (src/ggml-cpu/ggml-cpu.c:7428-7605) for Q4_0 / Q8_0 / Q4_K / Q5_K / Q6_K weights, in the CPU's arithmetic AND order
int ggml_cdna4_mul_mat_exact(int type, const void *W, int64_t w_row_bytes, const float *X, int64_t x_row_stride, float *Y, int64_t y_row_stride,
                             int64_t M, int64_t K, int64_t B, void *workspace, size_t workspace_bytes, void *stream) {
    NEEDX(ggml_cdna4_mul_mat_exact_supported(type, K), "mul_mat_exact: Q4_0 / Q8_0 weights with K a multiple of 32, or Q4_K / Q5_K / Q6_K with K a multiple of 256");
    if (M <= 0 || B <= 0) return 0;
    NEEDX(workspace && !((uintptr_t)workspace & 255) && workspace_bytes >= ggml_cdna4_mul_mat_exact_workspace_size(type, K, B), "mul_mat_exact: workspace too small or misaligned");
    NEEDX(!(((uintptr_t)X | (uintptr_t)(x_row_stride * 4)) & 15) && !(((uintptr_t)W | (uintptr_t)w_row_bytes) & 1), "mul_mat_exact: misaligned operands");
    int8_t *qs = (int8_t *)workspace;
    float *dd = (float *)((char *)workspace + exact_ws_d(K, B));
    const int64_t units = M * ((B + 7) / 8);
    const dim3 grid((unsigned)((units + 3) / 4));
    const hipStream_t st = (hipStream_t)stream;
    if (exact_kq(type)) {
        int16_t *bs = (int16_t *)((char *)workspace + exact_ws_bsums(type, K, B));
        int rc = cdna4_launch_quantize_q8_K(X, x_row_stride, K, B, qs, dd, bs, nullptr, st);                // quantize_row_q8_K_ref: what the CPU backend runs (ggml-cpu-quants.c quantize_row_q8_K)
        if (rc) return rc;
        if (type == CDNA4_Q4_K) hipLaunchKernelGGL(k_mul_mat_exact_kq<CDNA4_Q4_K>, grid, dim3(256), 0, st, (const uint8_t *)W, w_row_bytes, qs, dd, bs, Y, y_row_stride, (int)M, (int)K, (int)B);
        else if (type == CDNA4_Q5_K) hipLaunchKernelGGL(k_mul_mat_exact_kq<CDNA4_Q5_K>, grid, dim3(256), 0, st, (const uint8_t *)W, w_row_bytes, qs, dd, bs, Y, y_row_stride, (int)M, (int)K, (int)B);
        else hipLaunchKernelGGL(k_mul_mat_exact_kq<CDNA4_Q6_K>, grid, dim3(256), 0, st, (const uint8_t *)W, w_row_bytes, qs, dd, bs, Y, y_row_stride, (int)M, (int)K, (int)B);
        CDNA4_CHECK_LAUNCH();
        return 0;
    }
    int rc = cdna4_launch_quantize_q8_0(X, x_row_stride, K, B, qs, dd, nullptr, false, st);       // the AVX2 body of quantize_row_q8_0: what the CPU backend runs
    if (rc) return rc;
    if (type == CDNA4_Q4_0) hipLaunchKernelGGL(k_mul_mat_exact_q<CDNA4_Q4_0>, grid, dim3(256), 0, st, (const uint8_t *)W, w_row_bytes, qs, dd, Y, y_row_stride, (int)M, (int)K, (int)B);
    else hipLaunchKernelGGL(k_mul_mat_exact_q<CDNA4_Q8_0>, grid, dim3(256), 0, st, (const uint8_t *)W, w_row_bytes, qs, dd, Y, y_row_stride, (int)M, (int)K, (int)B);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int ggml_cdna4_op_mul_mat_f_exact(const T4 *a, const T4 *b, const T4 *d, void *stream) {
    NEEDX(a->type == CDNA4_F32 && b->type == CDNA4_F32 && d->type == CDNA4_F32, "mul_mat_f_exact: F32 x F32 only");
    NEEDX(a->ne[0] == b->ne[0] && d->ne[0] == a->ne[1] && d->ne[1] == b->ne[1] && d->ne[2] == b->ne[2] && d->ne[3] == b->ne[3], "mul_mat_f_exact: shape mismatch");
    NEEDX(a->ne[2] > 0 && a->ne[3] > 0 && b->ne[2] % a->ne[2] == 0 && b->ne[3] % a->ne[3] == 0, "mul_mat_f_exact: batch dims not broadcastable");
    NEEDX(a->nb[0] == 4 && b->nb[0] == 4, "mul_mat_f_exact: k must be the contiguous dimension");
    const int64_t nout = d->ne[0] * d->ne[1] * d->ne[2] * d->ne[3];
    if (nout <= 0) return 0;
    hipLaunchKernelGGL(k_mul_mat_exact_f32, dim3((unsigned)((nout + 7) / 8)), dim3(256), 0, (hipStream_t)stream, *a, *b, *d, nout);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int ggml_cdna4_op_norm_exact(const T4 *a, const T4 *d, float eps, void *stream) {
    NEEDX(a->type == CDNA4_F32 && d->type == CDNA4_F32 && a->nb[0] == 4 && d->nb[0] == 4, "norm_exact: F32 rows");
    NEEDX(a->ne[0] == d->ne[0] && a->ne[1] == d->ne[1] && a->ne[2] == d->ne[2] && a->ne[3] == d->ne[3], "norm_exact: shape mismatch");
    const int64_t nr = a->ne[1] * a->ne[2] * a->ne[3];
    if (nr <= 0 || a->ne[0] <= 0) return 0;
    hipLaunchKernelGGL(k_norm_exact, dim3((unsigned)((nr + 63) / 64)), dim3(64), 0, (hipStream_t)stream, *a, *d, eps, nr);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int ggml_cdna4_op_rms_norm_exact(const T4 *a, const T4 *d, float eps, void *stream) {
    NEEDX(a->type == CDNA4_F32 && d->type == CDNA4_F32 && a->nb[0] == 4 && d->nb[0] == 4, "rms_norm_exact: F32 rows");
    NEEDX(a->ne[0] == d->ne[0] && a->ne[1] == d->ne[1] && a->ne[2] == d->ne[2] && a->ne[3] == d->ne[3], "rms_norm_exact: shape mismatch");
    const int64_t nr = a->ne[1] * a->ne[2] * a->ne[3];
    if (nr <= 0 || a->ne[0] <= 0) return 0;
    hipLaunchKernelGGL(k_rms_norm_exact, dim3((unsigned)((nr + 63) / 64)), dim3(64), 0, (hipStream_t)stream, *a, *d, eps, nr);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
// contiguous F32 (what ggml_compute_forward_silu_f32 asserts: ggml-cpu.c:6369-6373)
int ggml_cdna4_op_silu_exact(const T4 *a, const T4 *d, void *stream) {
    NEEDX(a->type == CDNA4_F32 && d->type == CDNA4_F32, "silu_exact: F32");
    NEEDX(a->nb[0] == 4 && a->nb[1] == a->ne[0] * 4 && a->nb[2] == a->nb[1] * a->ne[1] && a->nb[3] == a->nb[2] * a->ne[2], "silu_exact: contiguous source");
    NEEDX(d->nb[0] == 4 && d->nb[1] == d->ne[0] * 4 && d->nb[2] == d->nb[1] * d->ne[1] && d->nb[3] == d->nb[2] * d->ne[2], "silu_exact: contiguous destination");
    NEEDX(a->ne[0] == d->ne[0] && a->ne[1] == d->ne[1] && a->ne[2] == d->ne[2] && a->ne[3] == d->ne[3], "silu_exact: shape mismatch");
    const int64_t n = a->ne[0] * a->ne[1] * a->ne[2] * a->ne[3];
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_silu_exact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float *)a->data, (float *)d->data, (int)a->ne[0], n);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
// contiguous rows, no mask, max_bias == 0 (what gpt-2's SCALE -> DIAG_MASK_INF -> SOFT_MAX chain leaves for the softmax node)
int ggml_cdna4_op_soft_max_exact(const T4 *a, const T4 *d, float scale, void *stream) {
    NEEDX(a->type == CDNA4_F32 && d->type == CDNA4_F32, "soft_max_exact: F32");
    NEEDX(a->nb[0] == 4 && a->nb[1] == a->ne[0] * 4 && a->nb[2] == a->nb[1] * a->ne[1] && a->nb[3] == a->nb[2] * a->ne[2], "soft_max_exact: contiguous source");
    NEEDX(d->nb[0] == 4 && d->nb[1] == d->ne[0] * 4 && d->nb[2] == d->nb[1] * d->ne[1] && d->nb[3] == d->nb[2] * d->ne[2] && d->ne[0] == a->ne[0], "soft_max_exact: contiguous destination");
    const int64_t nr = a->ne[1] * a->ne[2] * a->ne[3];
    if (nr <= 0 || a->ne[0] <= 0) return 0;
    hipLaunchKernelGGL(k_soft_max_exact, dim3((unsigned)((nr + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const float *)a->data, (float *)d->data, (int)a->ne[0], nr, scale);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
