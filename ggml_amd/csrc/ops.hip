// ops.hip — the supporting ops around MUL_MAT as plain HIP kernels (no LDS tricks, no MFMA: these are
// HBM/L2-bound elementwise / row-reduction ops, written for wave64 with coalesced 4-byte accesses).
// Semantics follow the reference CPU backend function by function (citations in include/ggml_cdna4.h and
// next to each kernel); compiled with -ffp-contract=off so mul-then-add sequences round like the CPU's.
#include "../../include/ggml_cdna4.h"
#include "cdna4_common.h"
#include "cdna4_kernels.h"
#include "epilogue.h"
#include "quantize_dev.h"
#include <math.h>

typedef ggml_cdna4_tensor T4;

static inline int64_t nelem(const T4 *t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
static inline int64_t nrows(const T4 *t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
static inline size_t tsize(int type) {
    switch (type) { case CDNA4_F32: case CDNA4_I32: return 4; case CDNA4_F16: case CDNA4_BF16: return 2; case CDNA4_Q4_0: return 18; case CDNA4_Q8_0: return 34;
                    case CDNA4_Q4_K: return 144; case CDNA4_Q5_K: return 176; case CDNA4_Q6_K: return 210;
                    case CDNA4_Q4_1: return 20; case CDNA4_Q5_0: return 22; case CDNA4_Q5_1: return 24; case CDNA4_Q2_K: return 84; case CDNA4_Q3_K: return 110;
                    case CDNA4_IQ4_NL: return 18; case CDNA4_IQ4_XS: return 136; }
    return 0;
}
static inline int bsize(int type) {
    switch (type) { case CDNA4_Q4_0: case CDNA4_Q8_0: case CDNA4_Q4_1: case CDNA4_Q5_0: case CDNA4_Q5_1: case CDNA4_IQ4_NL: return 32;
                    case CDNA4_Q4_K: case CDNA4_Q5_K: case CDNA4_Q6_K: case CDNA4_Q2_K: case CDNA4_Q3_K: case CDNA4_IQ4_XS: return 256; }
    return 1;
}
static inline bool is_contig(const T4 *t) {      // ggml_is_contiguous
    const int64_t bs = bsize(t->type);
    if (t->nb[0] != (int64_t)tsize(t->type)) return false;
    if (t->nb[1] != t->nb[0] * (t->ne[0] / bs)) return false;
    return t->nb[2] == t->nb[1] * t->ne[1] && t->nb[3] == t->nb[2] * t->ne[2];
}
static inline bool same_shape(const T4 *a, const T4 *b) { return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3]; }
static inline dim3 grid1d(int64_t n, int per_block = 256) { int64_t g = (n + per_block - 1) / per_block; if (g < 1) g = 1; return dim3((unsigned)g); }

struct idx4 { int64_t i0, i1, i2, i3; };
__device__ __forceinline__ idx4 unravel(int64_t i, const int64_t ne[4]) {
    idx4 r; r.i0 = i % ne[0]; i /= ne[0]; r.i1 = i % ne[1]; i /= ne[1]; r.i2 = i % ne[2]; r.i3 = i / ne[2]; return r;
}
__device__ __forceinline__ const char *at(const T4 &t, const idx4 &x) { return (const char *)t.data + x.i0 * t.nb[0] + x.i1 * t.nb[1] + x.i2 * t.nb[2] + x.i3 * t.nb[3]; }

__device__ __forceinline__ float block_sum(float v, float *red) {       // 256 threads
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float *red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------ binary
template <int OP>
__global__ __launch_bounds__(256) void k_binary(const T4 a, const T4 b, const T4 d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const idx4 x = unravel(i, d.ne);
    const idx4 y = {x.i0 % b.ne[0], x.i1 % b.ne[1], x.i2 % b.ne[2], x.i3 % b.ne[3]};
    const float u = *(const float *)at(a, x), v = *(const float *)at(b, y);
    float r;
    if (OP == GGML_CDNA4_ADD) r = u + v; else if (OP == GGML_CDNA4_SUB) r = u - v; else if (OP == GGML_CDNA4_MUL) r = u * v; else r = u / v;
    *(float *)at(d, x) = r;
}

// ------------------------------------------------------------------------------------------------ scale / unary
__global__ __launch_bounds__(256) void k_scale(const float *__restrict__ x, float *__restrict__ y, float s, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * s;
}
template <int OP>
__global__ __launch_bounds__(256) void k_unary(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    if (OP == GGML_CDNA4_GELU) {
        // the CPU reads a 64K-entry fp16 table indexed by fp16(x) (ggml_vec_gelu_f32, ggml-cpu.c:1759-1774)
        r = gelu_lut_f32(v);
    } else if (OP == GGML_CDNA4_GELU_QUICK) {
        if (v <= -10.0f || v >= 10.0f) { r = v * (1.0f / (1.0f + expf(-1.702f * v))); }
        else { const float xh = (float)(half_t)v; r = (float)(half_t)opaque_f32(xh * (1.0f / (1.0f + expf(-1.702f * xh)))); if (r == 0.0f) r = __builtin_copysignf(0.0f, xh); }
    } else if (OP == GGML_CDNA4_SILU) r = v / (1.0f + expf(-v));
    else if (OP == GGML_CDNA4_RELU) r = v > 0.f ? v : 0.f;
    else r = tanhf(v);
    y[i] = r;
}

// ------------------------------------------------------------------------------------------------ norm / rms_norm
// one 256-thread block per row; the row is re-read from L1/L2 (3 passes) — rows are K floats, a few KB
// gain / shift (nullptr = none): the MUL and ADD that follow a NORM in every transformer graph (gpt-2: main-backend.cpp:476-488),
// as the same two separate fp32 operations per element — bit-identical to NORM -> MUL -> ADD
// Q8K (round 5): the row that was just normalised is ALSO quantized to the fp16 GEMM image of its Q8_K rounding — the activations the next MUL_MAT multiplies — so that
// the graph's first reader of a normalised tensor pays no quantizer launch either.  The fp32 row is written as before (same code, same bits) and kept in LDS;
// sixteen lanes per 256-value block run q8_K_group16_image (quantize_dev.h: the arithmetic of k_quantize_q8_K, bit for bit) on it.  xh = the image [n / 128][B][128],
// pair-interleaved; n % 256 == 0, n <= 8192.
// IMG: 0 none, 1 the K-quants' image (Q8_K rounding, 256-value blocks, q8_K_group16_image), 2 the 32-block formats' image (Q8_0 rounding: the body of k_quantize_q8_0 —
// quantize_act.hip; q8_0_block<false>, eight lanes per block — n % 32 == 0)
template <bool RMS, int IMG = 0>
__global__ __launch_bounds__(256) void k_norm(const T4 a, const T4 d, float eps, const float *__restrict__ gain, const float *__restrict__ shift, uint8_t *__restrict__ xh, int64_t nrows_img) {
    __shared__ float red[4];
    constexpr bool Q8K = IMG == 1;
    __shared__ __attribute__((aligned(16))) float yrow[IMG ? 8192 : 4];
    const int64_t row = blockIdx.x;
    const idx4 x = {0, row % a.ne[1], (row / a.ne[1]) % a.ne[2], row / (a.ne[1] * a.ne[2])};
    const float *src = (const float *)at(a, x);
    float *dst = (float *)at(d, x);
    const int n = (int)a.ne[0];
    auto put = [&](int i, float v) { if (gain) v = v * gain[i]; if (shift) v = v + shift[i]; dst[i] = v; if constexpr (IMG != 0) yrow[i] = v; };
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = src[i]; s += RMS ? v * v : v; }
    s = block_sum(s, red);
    const float mean = s / n;
    if (RMS) {
        const float scale = 1.0f / sqrtf(mean + eps);
        for (int i = threadIdx.x; i < n; i += 256) put(i, src[i] * scale);
    } else {
        float s2 = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) { const float v = src[i] - mean; s2 += v * v; }
        s2 = block_sum(s2, red);
        const float scale = 1.0f / sqrtf(s2 / n + eps);
        for (int i = threadIdx.x; i < n; i += 256) put(i, (src[i] - mean) * scale);
    }
    if constexpr (Q8K) {
        __syncthreads();
        const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;          // sixteen 16-lane groups: block grp, grp + 16, ..
        for (int sb = grp; sb < n / 256; sb += 16) {
            float e[16];
#pragma unroll
            for (int i = 0; i < 8; i++) { e[i] = yrow[256 * sb + 8 * l16 + i]; e[8 + i] = yrow[256 * sb + 128 + 8 * l16 + i]; }
            u32x4 o0, o1;
            q8_K_group16_image(e, l16, o0, o1);
            uint8_t *p0 = xh + (((int64_t)(2 * sb) * nrows_img + row) * 128 + 8 * l16) * 2;
            *reinterpret_cast<u32x4 *>(p0) = o0;
            *reinterpret_cast<u32x4 *>(p0 + nrows_img * 256) = o1;
        }
    }
    if constexpr (IMG == 2) {
        __syncthreads();
        // one thread per four values, eight lanes per 32-block (whole groups drop out together: n % 32 == 0), exactly as k_quantize_q8_0<false> maps them
        for (int g4 = threadIdx.x; g4 < n / 4; g4 += 256) {
            const float e[4] = {yrow[4 * g4], yrow[4 * g4 + 1], yrow[4 * g4 + 2], yrow[4 * g4 + 3]};
            int q[4]; float dh;
            q8_0_block<false>(e, q, dh);
            half_t h[4];
#pragma unroll
            for (int i = 0; i < 4; i++) h[i] = (half_t)(dh * (float)q[i]);
            const int64_t k = 4 * (int64_t)g4;
            const half2_t lo = {h[0], h[2]}, hi = {h[1], h[3]};                 // pair-interleaved: (k0, k2, k1, k3)
            *reinterpret_cast<u32x2 *>(xh + (((k >> 7) * nrows_img + row) * 128 + (k & 127)) * 2) = u32x2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
        }
    }
}

// ------------------------------------------------------------------------------------------------ soft_max
template <int MASK>   // 0 none, 1 f32, 2 f16
__global__ __launch_bounds__(256) void k_soft_max(const float *__restrict__ x, const void *__restrict__ mask, float *__restrict__ y,
                                                  int nc, int ne01, int ne02, float scale, float max_bias, float m0, float m1, uint32_t n_head_log2,
                                                  float pre_scale, int use_pre, int diag_n_past) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float *sp = x + row * nc;
    float *dp = y + row * nc;
    const uint32_t hh = (uint32_t)((row / ne01) % ne02);
    const float slope = max_bias > 0.0f ? (hh < n_head_log2 ? powf(m0, (float)(hh + 1)) : powf(m1, (float)(2 * (hh - n_head_log2) + 1))) : 1.0f;
    const int64_t moff = (row % ne01) * (int64_t)nc;
    auto val = [&](int i) -> float {
        // use_pre / diag_n_past >= 0: the SCALE and DIAG_MASK_INF nodes in front of the softmax (gpt-2: main-backend.cpp:586-596), as their
        // own fp32 operations: (x * pre_scale), -inf right of the diagonal, then the softmax's own scale
        float v = sp[i];
        if (use_pre) v = v * pre_scale;
        if (diag_n_past >= 0 && i > diag_n_past + (int)(row % ne01)) v = -INFINITY;
        v = v * scale;
        if (MASK == 1) v += slope * ((const float *)mask)[moff + i];
        if (MASK == 2) v += slope * (float)((const half_t *)mask)[moff + i];
        return v;
    };
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < nc; i += 256) mx = fmaxf(mx, val(i));
    mx = block_max(mx, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < nc; i += 256) { const float e = expf(val(i) - mx); dp[i] = e; s += e; }
    s = block_sum(s, red);
    const float inv = 1.0f / s;
    for (int i = threadIdx.x; i < nc; i += 256) dp[i] *= inv;
}

// ------------------------------------------------------------------------------------------------ diag_mask_inf
__global__ __launch_bounds__(256) void k_diag_mask_inf(const float *__restrict__ x, float *__restrict__ y, int64_t n, int nc, int nr, int n_past) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % nc), row = (int)((i / nc) % nr);
    y[i] = col > n_past + row ? -INFINITY : x[i];
}

// ------------------------------------------------------------------------------------------------ dequantize element
// element k of a quantized row, with the exact operation order of dequantize_row_* (src/ggml-quants.c)
template <int TYPE> __device__ __forceinline__ float deq_elem(const uint8_t *row, int64_t k);
template <> __device__ __forceinline__ float deq_elem<CDNA4_F32>(const uint8_t *row, int64_t k) { return ((const float *)row)[k]; }
template <> __device__ __forceinline__ float deq_elem<CDNA4_F16>(const uint8_t *row, int64_t k) { return (float)((const half_t *)row)[k]; }
template <> __device__ __forceinline__ float deq_elem<CDNA4_BF16>(const uint8_t *row, int64_t k) {      // ggml_compute_bf16_to_fp32: the 16 bits are the upper half of the fp32
    const uint32_t b = (uint32_t)((const uint16_t *)row)[k] << 16; return __builtin_bit_cast(float, b);
}
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q4_0>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 5) * 18; const int j = (int)(k & 31);
    const float d = h2f(ld_u16(b)); const uint8_t q = b[2 + (j & 15)];
    return (float)((j < 16 ? (q & 0x0F) : (q >> 4)) - 8) * d;
}
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q8_0>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 5) * 34;
    return (float)((const int8_t *)b)[2 + (k & 31)] * h2f(ld_u16(b));
}
__device__ __forceinline__ void k4_sm(const uint8_t *q, int j, uint8_t &d, uint8_t &m) {
    if (j < 4) { d = q[j] & 63; m = q[j + 4] & 63; } else { d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q4_K>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 8) * 144; const int j = (int)(k & 255), g = j >> 6, l = j & 31, hi = (j >> 5) & 1;
    uint8_t sc, m; k4_sm(b + 4, 2 * g + hi, sc, m);
    const float d = h2f(ld_u16(b)) * sc, mn = h2f(ld_u16(b + 2)) * m;
    const uint8_t q = b[16 + 32 * g + l];
    return d * (float)(hi ? (q >> 4) : (q & 0xF)) - mn;
}
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q5_K>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 8) * 176; const int j = (int)(k & 255), g = j >> 6, l = j & 31, hi = (j >> 5) & 1;
    uint8_t sc, m; k4_sm(b + 4, 2 * g + hi, sc, m);
    const float d = h2f(ld_u16(b)) * sc, mn = h2f(ld_u16(b + 2)) * m;
    const uint8_t q = b[48 + 32 * g + l], h = b[16 + l];
    const int v = (hi ? (q >> 4) : (q & 0xF)) + ((h >> (2 * g + hi)) & 1 ? 16 : 0);
    return d * (float)v - mn;
}
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q6_K>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 8) * 210; const int j = (int)(k & 255), n = j >> 7, r = j & 127, qd = r >> 5, l = r & 31;
    const uint8_t ql = b[64 * n + l + ((qd & 1) ? 32 : 0)], qh = b[128 + 32 * n + l];
    const int q = (int)(int8_t)(((qd & 2) ? (ql >> 4) : (ql & 0xF)) | (((qh >> (2 * qd)) & 3) << 4)) - 32;
    const float d = h2f(ld_u16(b + 208));
    const int8_t sc = ((const int8_t *)b)[192 + 8 * n + (l >> 4) + 2 * qd];
    return d * sc * q;
}
// Q4_1 {fp16 d, fp16 m, qs[16]}: q * d + m (src/ggml-quants.c:275-293)
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q4_1>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 5) * 20; const int j = (int)(k & 31);
    const float d = h2f(ld_u16(b)), m = h2f(ld_u16(b + 2)); const uint8_t q = b[4 + (j & 15)];
    return (float)(j < 16 ? (q & 0x0F) : (q >> 4)) * d + m;
}
// IQ4_NL {fp16 d, qs[16]}: d * kvalues_iq4nl[code], codes laid out like Q4_0's nibbles (src/ggml-quants.c:2436-2452)
template <> __device__ __forceinline__ float deq_elem<CDNA4_IQ4_NL>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 5) * 18; const int j = (int)(k & 31);
    const float d = h2f(ld_u16(b)); const uint8_t q = b[2 + (j & 15)];
    return d * (float)(int8_t)(iq4nl_lut4(j < 16 ? (q & 0x0Fu) : (q >> 4)) & 0xFFu);
}
// IQ4_XS {fp16 d, u16 scales_h, scales_l[4], qs[128]}: eight IQ4_NL-shaped sub-blocks, dl = d * (ls - 32) with the 6-bit ls, dl * kvalues_iq4nl[code]
// (src/ggml-quants.c:2454-2475)
template <> __device__ __forceinline__ float deq_elem<CDNA4_IQ4_XS>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 8) * 136; const int e = (int)(k & 255), ib = e >> 5, j = e & 31;
    const uint32_t sh = ld_u16(b + 2);
    const int ls = (int)(((b[4 + (ib >> 1)] >> (4 * (ib & 1))) & 0xFu) | (((sh >> (2 * ib)) & 3u) << 4));
    const float dl = h2f(ld_u16(b)) * (float)(ls - 32);
    const uint8_t q = b[8 + 16 * ib + (j & 15)];
    return dl * (float)(int8_t)(iq4nl_lut4(j < 16 ? (q & 0x0Fu) : (q >> 4)) & 0xFFu);
}
// Q5_0 {fp16 d, qh[4], qs[16]}: bit j of qh is the fifth bit of weight j; (q - 16) * d (src/ggml-quants.c:295-319)
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q5_0>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 5) * 22; const int j = (int)(k & 31);
    const float d = h2f(ld_u16(b)); const uint32_t qh = ld_u32_a2(b + 2); const uint8_t q = b[6 + (j & 15)];
    return (float)((int)((j < 16 ? (q & 0x0F) : (q >> 4)) | (((qh >> j) & 1) << 4)) - 16) * d;
}
// Q5_1 {fp16 d, fp16 m, qh[4], qs[16]}: q * d + m (src/ggml-quants.c:321-347)
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q5_1>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 5) * 24; const int j = (int)(k & 31);
    const float d = h2f(ld_u16(b)), m = h2f(ld_u16(b + 2)); const uint32_t qh = ld_u32_a2(b + 4); const uint8_t q = b[8 + (j & 15)];
    return (float)((j < 16 ? (q & 0x0F) : (q >> 4)) | (((qh >> j) & 1) << 4)) * d + m;
}
// Q2_K {scales[16], qs[64], fp16 d, fp16 dmin}: weight 128 n + 32 sh + l is (qs[32 n + l] >> 2 sh) & 3 of sub-block 8 n + 2 sh + (l >> 4);
// d * (scale & 15) * q - dmin * (scale >> 4) (src/ggml-quants.c:712-744)
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q2_K>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 8) * 84; const int j = (int)(k & 255), n = j >> 7, sh = (j >> 5) & 3, l = j & 31;
    const uint8_t sc = b[8 * n + 2 * sh + (l >> 4)];
    const float dl = h2f(ld_u16(b + 80)) * (float)(sc & 0xF), ml = h2f(ld_u16(b + 82)) * (float)(sc >> 4);
    return dl * (float)((b[16 + 32 * n + l] >> (2 * sh)) & 3) - ml;
}
// Q3_K {hmask[32], qs[64], scales[12], fp16 d}: same walk; the third bit is bit 4 n + sh of hmask[l], CLEAR means -4; sixteen 6-bit scales
// (low 4 bits two per byte in scales[0..7], high 2 bits in scales[8..11]) minus 32 (src/ggml-quants.c:1056-1104)
template <> __device__ __forceinline__ float deq_elem<CDNA4_Q3_K>(const uint8_t *row, int64_t k) {
    const uint8_t *b = row + (k >> 8) * 110; const int j = (int)(k & 255), n = j >> 7, sh = (j >> 5) & 3, l = j & 31, g = 8 * n + 2 * sh + (l >> 4);
    const uint8_t *sc = b + 96;
    const int s6 = (g < 8 ? (sc[g] & 0xF) : (sc[g - 8] >> 4)) | (((sc[8 + (g & 3)] >> (2 * (g >> 2))) & 3) << 4);
    const float dl = h2f(ld_u16(b + 108)) * (float)(s6 - 32);
    const int q = (int)((b[32 + 32 * n + l] >> (2 * sh)) & 3) - ((b[l] >> (4 * n + sh)) & 1 ? 0 : 4);
    return dl * (float)q;
}

// ------------------------------------------------------------------------------------------------ get_rows
template <int TYPE>
__global__ __launch_bounds__(256) void k_get_rows(const T4 a, const T4 ids, const T4 d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const idx4 x = unravel(i, d.ne);                         // (k, i10, i11, i12)
    const int32_t r = *(const int32_t *)((const char *)ids.data + x.i1 * ids.nb[0] + x.i2 * ids.nb[1] + x.i3 * ids.nb[2]);
    const uint8_t *row = (const uint8_t *)a.data + (int64_t)r * a.nb[1] + x.i2 * a.nb[2] + x.i3 * a.nb[3];
    *(float *)at(d, x) = deq_elem<TYPE>(row, x.i0);
}

// ------------------------------------------------------------------------------------------------ cpy
// logical element i of src -> logical element i of dst (both enumerate ne[0] fastest), ggml_compute_forward_dup
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void k_cpy(const T4 a, const T4 d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = (float)*(const TS *)at(a, unravel(i, a.ne));
    *(TD *)at(d, unravel(i, d.ne)) = (TD)v;
}
template <int TYPE>
__global__ __launch_bounds__(256) void k_cpy_q_to_f32(const T4 a, const T4 d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const idx4 x = unravel(i, a.ne);
    const uint8_t *row = (const uint8_t *)a.data + x.i1 * a.nb[1] + x.i2 * a.nb[2] + x.i3 * a.nb[3];
    *(float *)at(d, unravel(i, d.ne)) = deq_elem<TYPE>(row, x.i0);
}
// the same rounded to fp16 into a dense buffer (element i of the logical order at dst[i]): a quantized K / V cache in front of FLASH_ATTN_EXT
template <int TYPE>
__global__ __launch_bounds__(256) void k_q_to_f16_dense(const T4 a, half_t *__restrict__ dst, int64_t n, int64_t dst_row) {      // dst rows dst_row >= ne[0] halves apart
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const idx4 x = unravel(i, a.ne);
    const uint8_t *row = (const uint8_t *)a.data + x.i1 * a.nb[1] + x.i2 * a.nb[2] + x.i3 * a.nb[3];
    float v = deq_elem<TYPE>(row, x.i0);
    // BF16 reaches beyond fp16's range: saturate at +-65504 instead of producing inf (a finite cache stays finite, as on the CPU; NaN stays NaN) — ADVICE r3
    if constexpr (TYPE == CDNA4_BF16) v = v != v ? v : fminf(fmaxf(v, -65504.f), 65504.f);
    dst[(i / a.ne[0]) * dst_row + x.i0] = (half_t)opaque_f32(v);     // fp16 of the fp32 VALUE to_float gives (two roundings, like the CPU)
}
// F32 -> Q4_0 / Q8_0: one thread per 32-block; src rows contiguous in ne[0], dst blocks enumerated in logical order
template <int TYPE, bool REF>
__global__ __launch_bounds__(256) void k_cpy_f32_to_q(const T4 a, const T4 d, int64_t nblocks) {
    const int64_t ib = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ib >= nblocks) return;
    const int64_t e0 = ib * 32;
    const float *src = (const float *)at(a, unravel(e0, a.ne));
    const int64_t bpr = d.ne[0] / 32;                                    // blocks per dst row
    const idx4 dr = {0, (ib / bpr) % d.ne[1], (ib / (bpr * d.ne[1])) % d.ne[2], ib / (bpr * d.ne[1] * d.ne[2])};
    uint8_t *out = (uint8_t *)d.data + dr.i1 * d.nb[1] + dr.i2 * d.nb[2] + dr.i3 * d.nb[3] + (ib % bpr) * (TYPE == CDNA4_Q8_0 ? 34 : 18);
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) v[j] = src[j];
    if (TYPE == CDNA4_Q8_0) {
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 32; j++) amax = fmaxf(amax, fabsf(v[j]));
        const float dd = amax / 127.f;
        const float id = REF ? (dd != 0.f ? 1.0f / dd : 0.f) : (amax != 0.f ? 127.f / amax : 0.f);
        *(uint16_t *)out = f2h_bits(dd);
#pragma unroll
        for (int j = 0; j < 32; j++) ((int8_t *)out)[2 + j] = (int8_t)(REF ? (int)roundf(v[j] * id) : (int)__builtin_rintf(v[j] * id));
    } else {                                                            // quantize_row_q4_0_ref, src/ggml-quants.c:31-66
        float amax = 0.f, mx = 0.f;
#pragma unroll
        for (int j = 0; j < 32; j++) if (amax < fabsf(v[j])) { amax = fabsf(v[j]); mx = v[j]; }
        // d = max / -8 (exact: a power of two); an all-zero block has d = -0.0 on the CPU (+0 / -8): the sign is flipped on the BITS so
        // that no floating-point simplification can drop it
        float dd = mx * -0.125f;
        // hipcc fuses the multiply into the fp16 conversion as v_fma_mixlo_f16(mx, -0.125, +0): for mx = 0 that is (-0) + (+0) = +0 and the
        // sign of the zero is gone (seen in the ISA; one byte of an all-zero block differed from the CPU's).  Keep the product opaque.
        asm volatile("" : "+v"(dd));
        const float id = dd != 0.f ? 1.0f / dd : 0.f;
        *(uint16_t *)out = f2h_bits(dd);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float x0 = v[j] * id, x1 = v[16 + j] * id;
            const int a0 = (int)(int8_t)(x0 + 8.5f), a1 = (int)(int8_t)(x1 + 8.5f);
            const uint8_t q0 = (uint8_t)(a0 < 15 ? a0 : 15), q1 = (uint8_t)(a1 < 15 ? a1 : 15);
            out[2 + j] = q0 | (q1 << 4);
        }
    }
}

// F32 -> Q4_1 / Q5_0 / Q5_1 (a quantized KV cache written by CPY): quantize_row_q4_1_ref / _q5_0_ref / _q5_1_ref, src/ggml-quants.c:68-193 — what
// type_traits_cpu[].from_float is for these types (ggml-cpu.c:271-296).  Same block walk as k_cpy_f32_to_q.
template <int TYPE>
__global__ __launch_bounds__(256) void k_cpy_f32_to_q45(const T4 a, const T4 d, int64_t nblocks) {
    constexpr int BS = TYPE == CDNA4_Q4_1 ? 20 : (TYPE == CDNA4_Q5_0 ? 22 : 24);
    constexpr bool MINMAX = TYPE != CDNA4_Q5_0, FIVE = TYPE != CDNA4_Q4_1;
    const int64_t ib = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ib >= nblocks) return;
    const int64_t e0 = ib * 32;
    const float *src = (const float *)at(a, unravel(e0, a.ne));
    const int64_t bpr = d.ne[0] / 32;
    const idx4 dr = {0, (ib / bpr) % d.ne[1], (ib / (bpr * d.ne[1])) % d.ne[2], ib / (bpr * d.ne[1] * d.ne[2])};
    uint8_t *out = (uint8_t *)d.data + dr.i1 * d.nb[1] + dr.i2 * d.nb[2] + dr.i3 * d.nb[3] + (ib % bpr) * BS;
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) v[j] = src[j];
    float dd, sub, add;                                                 // q = (int8)((x - sub) * id + add), clamped
    if (MINMAX) {
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
#pragma unroll
        for (int j = 0; j < 32; j++) { if (v[j] < mn) mn = v[j]; if (v[j] > mx) mx = v[j]; }
        dd = (mx - mn) / (float)(FIVE ? 31 : 15); sub = mn; add = 0.5f;
        *(uint16_t *)(out + 2) = f2h_bits(mn);
    } else {
        float amax = 0.f, mx = 0.f;
#pragma unroll
        for (int j = 0; j < 32; j++) if (amax < fabsf(v[j])) { amax = fabsf(v[j]); mx = v[j]; }
        dd = mx * -0.0625f;                                              // max / -16, exact; the sign of a zero kept on the bits (see k_cpy_f32_to_q)
        asm volatile("" : "+v"(dd));
        sub = 0.f; add = 16.5f;
    }
    const float id = dd != 0.f ? 1.0f / dd : 0.f;
    *(uint16_t *)out = f2h_bits(dd);
    uint32_t qh = 0;
    uint8_t *qs = out + (FIVE ? (MINMAX ? 8 : 6) : 4);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float x0 = MINMAX ? (v[j] - sub) * id : v[j] * id, x1 = MINMAX ? (v[16 + j] - sub) * id : v[16 + j] * id;
        int a0, a1;
        if (TYPE == CDNA4_Q5_1) { a0 = (int)(uint8_t)(x0 + add); a1 = (int)(uint8_t)(x1 + add); }              // (uint8_t)(x + 0.5f), no clamp (:176-177)
        else { a0 = (int)(int8_t)(x0 + add); a1 = (int)(int8_t)(x1 + add); const int top = FIVE ? 31 : 15; a0 = a0 < top ? a0 : top; a1 = a1 < top ? a1 : top; }
        qs[j] = (uint8_t)((a0 & 0x0F) | ((a1 & 0x0F) << 4));
        if (FIVE) { qh |= (uint32_t)((a0 & 0x10) >> 4) << j; qh |= (uint32_t)((a1 & 0x10) >> 4) << (j + 16); }
    }
    if (FIVE) { uint8_t *ph = out + (MINMAX ? 4 : 2); *(uint16_t *)ph = (uint16_t)qh; *(uint16_t *)(ph + 2) = (uint16_t)(qh >> 16); }
}

// ------------------------------------------------------------------------------------------------ mul_mat F32/F16
// one wave per output element (m, n, batch): lanes stride over k (both operands contiguous in k)
template <typename TW>
__global__ __launch_bounds__(256) void k_mul_mat_f(const T4 a, const T4 b, const T4 d, int64_t nout) {
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= nout) return;
    const int lane = threadIdx.x & 63;
    const idx4 x = unravel(o, d.ne);                                   // (m, n, i2, i3)
    const int64_t r2 = b.ne[2] / a.ne[2], r3 = b.ne[3] / a.ne[3];
    const TW *w = (const TW *)((const char *)a.data + x.i0 * a.nb[1] + (x.i2 / r2) * a.nb[2] + (x.i3 / r3) * a.nb[3]);
    const float *v = (const float *)((const char *)b.data + x.i1 * b.nb[1] + x.i2 * b.nb[2] + x.i3 * b.nb[3]);
    float s = 0.f;
    const int K = (int)a.ne[0];
    for (int k = lane; k < K; k += 64) {
        float xv = v[k];
        if (sizeof(TW) == 2) xv = (float)(half_t)xv;                  // vec_dot_type F16: src1 is rounded to fp16 first
        s = fmaf((float)w[k], xv, s);
    }
    s = wave_sum(s);
    if (lane == 0) *(float *)at(d, x) = s;
}

// ------------------------------------------------------------------------------------------------ rope
__device__ __forceinline__ float yarn_ramp(float low, float high, int i0) {
    const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low);
    return 1 - fminf(1, fmaxf(0, y));
}
__global__ __launch_bounds__(256) void k_rope(const T4 a, const int32_t *__restrict__ pos, const float *__restrict__ ff, const T4 d, int64_t npairs,
                                              int n_dims, int neox, float theta_scale, float freq_scale, float ext_factor, float attn_factor, float c0, float c1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const int64_t hp = a.ne[0] / 2;
    const int64_t ip = i % hp; int64_t r = i / hp;
    const int64_t i1 = r % a.ne[1]; r /= a.ne[1];
    const int64_t i2 = r % a.ne[2], i3 = r / a.ne[2];
    const int64_t i0 = 2 * ip;
    const char *sb = (const char *)a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3];
    char *db = (char *)d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3];
    if (i0 >= n_dims) {                                               // untouched tail channels
        *(float *)(db + i0 * d.nb[0]) = *(const float *)(sb + i0 * a.nb[0]);
        *(float *)(db + (i0 + 1) * d.nb[0]) = *(const float *)(sb + (i0 + 1) * a.nb[0]);
        return;
    }
    float theta = (float)pos[i2];
    for (int64_t t = 0; t < ip; t++) theta *= theta_scale;           // same recurrence as ggml_rope_cache_init
    const float fq = ff ? ff[ip] : 1.0f;
    const float theta_extrap = theta / fq;
    const float theta_interp = freq_scale * theta_extrap;
    float th = theta_interp, mscale = attn_factor;
    if (ext_factor != 0.0f) {
        const float ramp_mix = yarn_ramp(c0, c1, (int)i0) * ext_factor;
        th = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
        mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
    }
    const float ct = cosf(th) * mscale, st = sinf(th) * mscale;
    const int64_t ia = neox ? ip : i0, ib = neox ? ip + n_dims / 2 : i0 + 1;
    const float x0 = *(const float *)(sb + ia * a.nb[0]), x1 = *(const float *)(sb + ib * a.nb[0]);
    *(float *)(db + ia * d.nb[0]) = x0 * ct - x1 * st;
    *(float *)(db + ib * d.nb[0]) = x0 * st + x1 * ct;
}

// ============================================================================================================
#define NEED(cond, msg) do { if (!(cond)) return cdna4_set_error_msg(msg); } while (0)

// ------------------------------------------------------------------------------------------------ MUL_MAT tail (GEMM path)
__global__ __launch_bounds__(256) void k_epilogue(float *__restrict__ Y, int64_t y_row_stride, int64_t M, int64_t n, const cdna4_epilogue e) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / M, m = i % M;
    Y[b * y_row_stride + m] = epilogue_apply(e, Y[b * y_row_stride + m], m, b);
}
int cdna4_launch_epilogue(float *Y, int64_t y_row_stride, int64_t M, int64_t B, const cdna4_epilogue &e, hipStream_t st) {
    if (M <= 0 || B <= 0 || (!e.bias && !e.resid && !e.act)) return 0;
    hipLaunchKernelGGL(k_epilogue, grid1d(M * B), dim3(256), 0, st, Y, y_row_stride, M, M * B, e);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

extern "C" {

int ggml_cdna4_op_binary(int op, const T4 *a, const T4 *b, const T4 *d, void *stream) {
    NEED(a->type == CDNA4_F32 && b->type == CDNA4_F32 && d->type == CDNA4_F32, "binary: F32 only");
    NEED(same_shape(a, d), "binary: src0 and dst shapes differ");
    for (int i = 0; i < 4; i++) NEED(b->ne[i] > 0 && a->ne[i] % b->ne[i] == 0, "binary: src1 is not broadcastable");
    const int64_t n = nelem(d);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    switch (op) {
        case GGML_CDNA4_ADD: hipLaunchKernelGGL(k_binary<GGML_CDNA4_ADD>, grid1d(n), dim3(256), 0, st, *a, *b, *d, n); break;
        case GGML_CDNA4_SUB: hipLaunchKernelGGL(k_binary<GGML_CDNA4_SUB>, grid1d(n), dim3(256), 0, st, *a, *b, *d, n); break;
        case GGML_CDNA4_MUL: hipLaunchKernelGGL(k_binary<GGML_CDNA4_MUL>, grid1d(n), dim3(256), 0, st, *a, *b, *d, n); break;
        case GGML_CDNA4_DIV: hipLaunchKernelGGL(k_binary<GGML_CDNA4_DIV>, grid1d(n), dim3(256), 0, st, *a, *b, *d, n); break;
        default: return cdna4_set_error_msg("binary: unknown op");
    }
    CDNA4_CHECK_LAUNCH();
    return 0;
}

// dst[i] = parts[0][i] + parts[1][i] + .. in THAT order (deterministic): the reduction behind a K-split MUL_MAT whose shards share a device, and the fall-back
// of the plug-in's RCCL all-reduce (ggml_cdna4_split.cpp).  Up to 16 parts; dst may be parts[0].
struct sum_parts_t { const float *p[16]; };
__global__ __launch_bounds__(256) void k_sum_partials(float *__restrict__ dst, const sum_parts_t parts, int n, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (4 * i + 3 < count) {
        float4 s = reinterpret_cast<const float4 *>(parts.p[0])[i];
        for (int k = 1; k < n; k++) { const float4 v = reinterpret_cast<const float4 *>(parts.p[k])[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4 *>(dst)[i] = s;
    } else for (int64_t e = 4 * i; e < count; e++) {                    // (the last, partial quad)
        float s = parts.p[0][e];
        for (int k = 1; k < n; k++) s += parts.p[k][e];
        dst[e] = s;
    }
}
int ggml_cdna4_sum_partials(float *dst, const float *const *parts, int n, int64_t count, void *stream) {
    NEED(n >= 1 && n <= 16 && count >= 0, "sum_partials: 1 .. 16 parts");
    NEED((((uintptr_t)dst) & 15) == 0, "sum_partials: pointers must be 16-byte aligned");
    sum_parts_t sp{};
    for (int k = 0; k < n; k++) { NEED(parts[k] && (((uintptr_t)parts[k]) & 15) == 0, "sum_partials: pointers must be 16-byte aligned"); sp.p[k] = parts[k]; }
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_sum_partials, grid1d((count + 3) / 4), dim3(256), 0, (hipStream_t)stream, dst, sp, n, count);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

int ggml_cdna4_op_scale(const T4 *a, const T4 *d, float scale, void *stream) {
    NEED(a->type == CDNA4_F32 && d->type == CDNA4_F32 && is_contig(a) && is_contig(d) && nelem(a) == nelem(d), "scale: contiguous F32 only");
    const int64_t n = nelem(d);
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_scale, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const float *)a->data, (float *)d->data, scale, n);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

int ggml_cdna4_op_unary(int op, const T4 *a, const T4 *d, void *stream) {
    NEED(a->type == CDNA4_F32 && d->type == CDNA4_F32 && is_contig(a) && is_contig(d) && nelem(a) == nelem(d), "unary: contiguous F32 only");
    const int64_t n = nelem(d);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream; const float *x = (const float *)a->data; float *y = (float *)d->data;
    switch (op) {
        case GGML_CDNA4_GELU: hipLaunchKernelGGL(k_unary<GGML_CDNA4_GELU>, grid1d(n), dim3(256), 0, st, x, y, n); break;
        case GGML_CDNA4_GELU_QUICK: hipLaunchKernelGGL(k_unary<GGML_CDNA4_GELU_QUICK>, grid1d(n), dim3(256), 0, st, x, y, n); break;
        case GGML_CDNA4_SILU: hipLaunchKernelGGL(k_unary<GGML_CDNA4_SILU>, grid1d(n), dim3(256), 0, st, x, y, n); break;
        case GGML_CDNA4_RELU: hipLaunchKernelGGL(k_unary<GGML_CDNA4_RELU>, grid1d(n), dim3(256), 0, st, x, y, n); break;
        case GGML_CDNA4_TANH: hipLaunchKernelGGL(k_unary<GGML_CDNA4_TANH>, grid1d(n), dim3(256), 0, st, x, y, n); break;
        default: return cdna4_set_error_msg("unary: unknown op");
    }
    CDNA4_CHECK_LAUNCH();
    return 0;
}

int ggml_cdna4_op_norm_affine(const T4 *a, const T4 *gain, const T4 *shift, const T4 *d, float eps, int rms, void *stream) {
    NEED(a->type == CDNA4_F32 && d->type == CDNA4_F32 && same_shape(a, d) && a->nb[0] == 4 && d->nb[0] == 4, "norm: F32 rows only");
    for (const T4 *g : {gain, shift})
        NEED(!g || (g->type == CDNA4_F32 && g->nb[0] == 4 && g->ne[0] == a->ne[0] && g->ne[1] == 1 && g->ne[2] == 1 && g->ne[3] == 1), "norm: gain / shift are F32 vectors of ne[0] elements");
    const int64_t nr = nrows(a);
    if (nr == 0 || a->ne[0] == 0) return 0;
    const float *gp = gain ? (const float *)gain->data : nullptr, *sp = shift ? (const float *)shift->data : nullptr;
    if (rms) hipLaunchKernelGGL((k_norm<true, 0>), dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream, *a, *d, eps, gp, sp, (uint8_t *)nullptr, (int64_t)0);
    else hipLaunchKernelGGL((k_norm<false, 0>), dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream, *a, *d, eps, gp, sp, (uint8_t *)nullptr, (int64_t)0);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
// the same + the Q8_K-rounded fp16 GEMM image of dst's rows into xh (capi.hip: ggml_cdna4_op_norm_affine_q8_K carves it out of a MUL_MAT workspace)
// kq != 0: the Q8_K image (rows of whole 256-value superblocks); 0: the Q8_0 image (rows of whole 32-value blocks)
int cdna4_launch_norm_affine_q8_K(const T4 *a, const T4 *gain, const T4 *shift, const T4 *d, float eps, int rms, void *xh, void *stream, int kq) {
    NEED(a->type == CDNA4_F32 && d->type == CDNA4_F32 && same_shape(a, d) && a->nb[0] == 4 && d->nb[0] == 4, "norm: F32 rows only");
    for (const T4 *g : {gain, shift})
        NEED(!g || (g->type == CDNA4_F32 && g->nb[0] == 4 && g->ne[0] == a->ne[0] && g->ne[1] == 1 && g->ne[2] == 1 && g->ne[3] == 1), "norm: gain / shift are F32 vectors of ne[0] elements");
    NEED(a->ne[0] % (kq ? 256 : 32) == 0 && a->ne[0] <= 8192 && a->ne[0] > 0, "norm + activation image: rows of up to 8192 values, whole blocks");
    NEED(xh && !((uintptr_t)xh & 15), "norm_q8_K: the image must be 16-byte aligned");
    const int64_t nr = nrows(a);
    if (nr == 0) return 0;
    const float *gp = gain ? (const float *)gain->data : nullptr, *sp = shift ? (const float *)shift->data : nullptr;
    if (kq) {
        if (rms) hipLaunchKernelGGL((k_norm<true, 1>), dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream, *a, *d, eps, gp, sp, (uint8_t *)xh, nr);
        else hipLaunchKernelGGL((k_norm<false, 1>), dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream, *a, *d, eps, gp, sp, (uint8_t *)xh, nr);
    } else {
        if (rms) hipLaunchKernelGGL((k_norm<true, 2>), dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream, *a, *d, eps, gp, sp, (uint8_t *)xh, nr);
        else hipLaunchKernelGGL((k_norm<false, 2>), dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream, *a, *d, eps, gp, sp, (uint8_t *)xh, nr);
    }
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int ggml_cdna4_op_norm(const T4 *a, const T4 *d, float eps, int rms, void *stream) { return ggml_cdna4_op_norm_affine(a, nullptr, nullptr, d, eps, rms, stream); }

int ggml_cdna4_op_soft_max_ext(const T4 *a, const T4 *mask, const T4 *d, float scale, float max_bias, int use_pre_scale, float pre_scale, int diag_n_past, void *stream) {
    NEED(a->type == CDNA4_F32 && d->type == CDNA4_F32 && is_contig(a) && is_contig(d) && same_shape(a, d), "soft_max: contiguous F32 only");
    int mt = 0;
    if (mask) {
        NEED((mask->type == CDNA4_F32 || mask->type == CDNA4_F16) && is_contig(mask) && mask->ne[0] == a->ne[0] && mask->ne[1] >= a->ne[1], "soft_max: bad mask");
        mt = mask->type == CDNA4_F32 ? 1 : 2;
    }
    const int64_t nr = nrows(a);
    if (nr == 0 || a->ne[0] == 0) return 0;
    const uint32_t n_head = (uint32_t)a->ne[2];
    const uint32_t n_head_log2 = 1u << (uint32_t)floor(log2((double)n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    hipStream_t st = (hipStream_t)stream;
    const float *x = (const float *)a->data; float *y = (float *)d->data; const void *mp = mask ? mask->data : nullptr;
    const int nc = (int)a->ne[0], ne01 = (int)a->ne[1], ne02 = (int)a->ne[2];
    if (mt == 0) hipLaunchKernelGGL(k_soft_max<0>, dim3((unsigned)nr), dim3(256), 0, st, x, mp, y, nc, ne01, ne02, scale, max_bias, m0, m1, n_head_log2, pre_scale, use_pre_scale, diag_n_past);
    else if (mt == 1) hipLaunchKernelGGL(k_soft_max<1>, dim3((unsigned)nr), dim3(256), 0, st, x, mp, y, nc, ne01, ne02, scale, max_bias, m0, m1, n_head_log2, pre_scale, use_pre_scale, diag_n_past);
    else hipLaunchKernelGGL(k_soft_max<2>, dim3((unsigned)nr), dim3(256), 0, st, x, mp, y, nc, ne01, ne02, scale, max_bias, m0, m1, n_head_log2, pre_scale, use_pre_scale, diag_n_past);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int ggml_cdna4_op_soft_max(const T4 *a, const T4 *mask, const T4 *d, float scale, float max_bias, void *stream) {
    return ggml_cdna4_op_soft_max_ext(a, mask, d, scale, max_bias, 0, 1.0f, -1, stream);
}


int ggml_cdna4_op_diag_mask_inf(const T4 *a, const T4 *d, int n_past, void *stream) {
    NEED(a->type == CDNA4_F32 && d->type == CDNA4_F32 && is_contig(a) && is_contig(d) && same_shape(a, d) && n_past >= 0, "diag_mask_inf: contiguous F32 only");
    const int64_t n = nelem(d);
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_diag_mask_inf, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const float *)a->data, (float *)d->data, n, (int)a->ne[0], (int)a->ne[1], n_past);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

int ggml_cdna4_op_get_rows(const T4 *a, const T4 *ids, const T4 *d, void *stream) {
    NEED(ids->type == CDNA4_I32 && d->type == CDNA4_F32, "get_rows: ids must be I32 and dst F32");
    NEED(d->ne[0] == a->ne[0] && d->ne[1] == ids->ne[0] && d->ne[2] == ids->ne[1] && d->ne[3] == ids->ne[2], "get_rows: shape mismatch");
    NEED(a->ne[2] == ids->ne[1] && a->ne[3] == ids->ne[2] && a->nb[0] == (int64_t)tsize(a->type), "get_rows: batch dims mismatch");
    const int64_t n = nelem(d);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#define GR(T) hipLaunchKernelGGL(k_get_rows<T>, grid1d(n), dim3(256), 0, st, *a, *ids, *d, n); break
    switch (a->type) {
        case CDNA4_F32: GR(CDNA4_F32); case CDNA4_F16: GR(CDNA4_F16); case CDNA4_Q4_0: GR(CDNA4_Q4_0); case CDNA4_Q8_0: GR(CDNA4_Q8_0);
        case CDNA4_Q4_K: GR(CDNA4_Q4_K); case CDNA4_Q5_K: GR(CDNA4_Q5_K); case CDNA4_Q6_K: GR(CDNA4_Q6_K);
        case CDNA4_Q4_1: GR(CDNA4_Q4_1); case CDNA4_Q5_0: GR(CDNA4_Q5_0); case CDNA4_Q5_1: GR(CDNA4_Q5_1); case CDNA4_Q2_K: GR(CDNA4_Q2_K); case CDNA4_Q3_K: GR(CDNA4_Q3_K); case CDNA4_IQ4_NL: GR(CDNA4_IQ4_NL); case CDNA4_IQ4_XS: GR(CDNA4_IQ4_XS);
        default: return cdna4_set_error_msg("get_rows: unsupported source type");
    }
#undef GR
    CDNA4_CHECK_LAUNCH();
    return 0;
}

int ggml_cdna4_op_cpy(const T4 *a, const T4 *d, int q8_0_ref_rounding, void *stream) {
    const int64_t n = nelem(a);
    NEED(n == nelem(d), "cpy: element counts differ");
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ta = a->type, td = d->type;
    if ((ta == CDNA4_F32 || ta == CDNA4_F16) && (td == CDNA4_F32 || td == CDNA4_F16)) {
        if (ta == td && is_contig(a) && is_contig(d)) {
            hipError_t e = hipMemcpyAsync(d->data, a->data, (size_t)n * tsize(ta), hipMemcpyDeviceToDevice, st);
            return e == hipSuccess ? 0 : cdna4_set_error(e, __FILE__, __LINE__);
        }
        if (ta == CDNA4_F32 && td == CDNA4_F32) hipLaunchKernelGGL((k_cpy<float, float>), grid1d(n), dim3(256), 0, st, *a, *d, n);
        else if (ta == CDNA4_F32) hipLaunchKernelGGL((k_cpy<float, half_t>), grid1d(n), dim3(256), 0, st, *a, *d, n);
        else if (td == CDNA4_F32) hipLaunchKernelGGL((k_cpy<half_t, float>), grid1d(n), dim3(256), 0, st, *a, *d, n);
        else hipLaunchKernelGGL((k_cpy<half_t, half_t>), grid1d(n), dim3(256), 0, st, *a, *d, n);
    } else if (td == CDNA4_F32 && bsize(ta) > 1) {
        NEED(a->nb[0] == (int64_t)tsize(ta), "cpy: quantized source rows must be contiguous");
#define CQ(T) hipLaunchKernelGGL(k_cpy_q_to_f32<T>, grid1d(n), dim3(256), 0, st, *a, *d, n); break
        switch (ta) { case CDNA4_Q4_0: CQ(CDNA4_Q4_0); case CDNA4_Q8_0: CQ(CDNA4_Q8_0); case CDNA4_Q4_K: CQ(CDNA4_Q4_K); case CDNA4_Q5_K: CQ(CDNA4_Q5_K); case CDNA4_Q6_K: CQ(CDNA4_Q6_K);
                      case CDNA4_Q4_1: CQ(CDNA4_Q4_1); case CDNA4_Q5_0: CQ(CDNA4_Q5_0); case CDNA4_Q5_1: CQ(CDNA4_Q5_1); case CDNA4_Q2_K: CQ(CDNA4_Q2_K); case CDNA4_Q3_K: CQ(CDNA4_Q3_K); case CDNA4_IQ4_NL: CQ(CDNA4_IQ4_NL); case CDNA4_IQ4_XS: CQ(CDNA4_IQ4_XS);
                      default: return cdna4_set_error_msg("cpy: unsupported quantized source"); }
#undef CQ
    } else if (ta == CDNA4_F32 && (td == CDNA4_Q8_0 || td == CDNA4_Q4_0)) {
        NEED(a->nb[0] == 4 && a->ne[0] % 32 == 0 && d->ne[0] % 32 == 0 && d->nb[0] == (int64_t)tsize(td), "cpy: f32->q needs whole 32-blocks per row");
        const int64_t nbk = n / 32;
        if (td == CDNA4_Q4_0) hipLaunchKernelGGL((k_cpy_f32_to_q<CDNA4_Q4_0, true>), grid1d(nbk), dim3(256), 0, st, *a, *d, nbk);
        else if (q8_0_ref_rounding) hipLaunchKernelGGL((k_cpy_f32_to_q<CDNA4_Q8_0, true>), grid1d(nbk), dim3(256), 0, st, *a, *d, nbk);
        else hipLaunchKernelGGL((k_cpy_f32_to_q<CDNA4_Q8_0, false>), grid1d(nbk), dim3(256), 0, st, *a, *d, nbk);
    } else if (ta == CDNA4_F32 && (td == CDNA4_Q4_1 || td == CDNA4_Q5_0 || td == CDNA4_Q5_1)) {
        NEED(a->nb[0] == 4 && a->ne[0] % 32 == 0 && d->ne[0] % 32 == 0 && d->nb[0] == (int64_t)tsize(td), "cpy: f32->q needs whole 32-blocks per row");
        const int64_t nbk = n / 32;
        if (td == CDNA4_Q4_1) hipLaunchKernelGGL(k_cpy_f32_to_q45<CDNA4_Q4_1>, grid1d(nbk), dim3(256), 0, st, *a, *d, nbk);
        else if (td == CDNA4_Q5_0) hipLaunchKernelGGL(k_cpy_f32_to_q45<CDNA4_Q5_0>, grid1d(nbk), dim3(256), 0, st, *a, *d, nbk);
        else hipLaunchKernelGGL(k_cpy_f32_to_q45<CDNA4_Q5_1>, grid1d(nbk), dim3(256), 0, st, *a, *d, nbk);
    } else return cdna4_set_error_msg("cpy: unsupported type pair");
    CDNA4_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
bool cdna4_to_f16_dense_supported(int type) { return type == CDNA4_BF16 || type == CDNA4_Q4_0 || type == CDNA4_Q4_1 || type == CDNA4_Q5_0 || type == CDNA4_Q5_1 || type == CDNA4_Q8_0; }
int cdna4_launch_to_f16_dense(const T4 *a, void *dst, int64_t dst_row, hipStream_t st) {
    NEED(cdna4_to_f16_dense_supported(a->type), "to_f16_dense: unsupported source type");
    NEED(a->nb[0] == (int64_t)tsize(a->type) && a->ne[0] % bsize(a->type) == 0, "to_f16_dense: quantized source rows must be contiguous");
    const int64_t n = nelem(a);
    if (n == 0) return 0;
    NEED(dst_row >= a->ne[0], "to_f16_dense: destination rows overlap");
#define TF(T) hipLaunchKernelGGL(k_q_to_f16_dense<T>, grid1d(n), dim3(256), 0, st, *a, (half_t *)dst, n, dst_row); break
    switch (a->type) { case CDNA4_BF16: TF(CDNA4_BF16); case CDNA4_Q4_0: TF(CDNA4_Q4_0); case CDNA4_Q4_1: TF(CDNA4_Q4_1); case CDNA4_Q5_0: TF(CDNA4_Q5_0); case CDNA4_Q5_1: TF(CDNA4_Q5_1); default: TF(CDNA4_Q8_0); }
#undef TF
    CDNA4_CHECK_LAUNCH();
    return 0;
}
extern "C" {

int ggml_cdna4_op_mul_mat_f(const T4 *a, const T4 *b, const T4 *d, void *stream) {
    NEED((a->type == CDNA4_F32 || a->type == CDNA4_F16) && b->type == CDNA4_F32 && d->type == CDNA4_F32, "mul_mat_f: F32/F16 x F32 only");
    NEED(a->ne[0] == b->ne[0] && d->ne[0] == a->ne[1] && d->ne[1] == b->ne[1] && d->ne[2] == b->ne[2] && d->ne[3] == b->ne[3], "mul_mat_f: shape mismatch");
    NEED(a->ne[2] > 0 && a->ne[3] > 0 && b->ne[2] % a->ne[2] == 0 && b->ne[3] % a->ne[3] == 0, "mul_mat_f: batch dims not broadcastable");
    NEED(a->nb[0] == (int64_t)tsize(a->type) && b->nb[0] == 4, "mul_mat_f: k must be the contiguous dimension");
    const int64_t nout = nelem(d);
    if (nout == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (a->type == CDNA4_F32) hipLaunchKernelGGL(k_mul_mat_f<float>, grid1d(nout, 4), dim3(256), 0, st, *a, *b, *d, nout);
    else hipLaunchKernelGGL(k_mul_mat_f<half_t>, grid1d(nout, 4), dim3(256), 0, st, *a, *b, *d, nout);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

int ggml_cdna4_op_rope(const T4 *a, const T4 *pos, const T4 *ffac, const T4 *d, int n_dims, int mode, int n_ctx_orig,
                       float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow, void *stream) {
    NEED(a->type == CDNA4_F32 && d->type == CDNA4_F32 && same_shape(a, d) && pos->type == CDNA4_I32, "rope: F32 data, I32 positions");
    NEED((mode & ~2) == 0, "rope: only NORMAL and NEOX modes");
    NEED(n_dims % 2 == 0 && n_dims <= a->ne[0] && a->ne[0] % 2 == 0 && pos->ne[0] >= a->ne[2], "rope: bad n_dims / positions");
    if (ffac) NEED(ffac->type == CDNA4_F32 && ffac->ne[0] >= n_dims / 2, "rope: bad freq_factors");
    const int64_t npairs = nelem(a) / 2;
    if (npairs == 0) return 0;
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    // ggml_rope_yarn_corr_dims, src/ggml.c:3699-3707
    auto corr_dim = [&](float n_rot) { return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float)M_PI)) / (2 * logf(freq_base)); };
    const float c0 = fmaxf(0.f, floorf(corr_dim(beta_fast))), c1 = fminf((float)(n_dims - 1), ceilf(corr_dim(beta_slow)));
    hipLaunchKernelGGL(k_rope, grid1d(npairs), dim3(256), 0, (hipStream_t)stream, *a, (const int32_t *)pos->data, ffac ? (const float *)ffac->data : nullptr, *d, npairs,
                       n_dims, (mode & 2) ? 1 : 0, theta_scale, freq_scale, ext_factor, attn_factor, c0, c1);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

int ggml_cdna4_dequantize_row(int type, const void *x, float *y, int64_t k, void *stream) {
    NEED(bsize(type) > 1 && k % bsize(type) == 0, "dequantize_row: unsupported type or ragged k");
    T4 a{}, d{};
    a.data = (void *)x; a.type = type; a.ne[0] = k; a.ne[1] = a.ne[2] = a.ne[3] = 1;
    a.nb[0] = (int64_t)tsize(type); a.nb[1] = a.nb[2] = a.nb[3] = (k / bsize(type)) * a.nb[0];
    d.data = y; d.type = CDNA4_F32; d.ne[0] = k; d.ne[1] = d.ne[2] = d.ne[3] = 1; d.nb[0] = 4; d.nb[1] = d.nb[2] = d.nb[3] = 4 * k;
    return ggml_cdna4_op_cpy(&a, &d, 0, stream);
}

}  // extern "C"
