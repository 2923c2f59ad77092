/*
 * oracle/ggml_oracle.c — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product
 * (ggml_amd/, the HIP kernels, the backend plug-in).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and there only as the checker.
 *
 * A plain-C restatement of the reference's CPU algorithm for the quantized MUL_MAT hot path
 * (ggerganov/ggml @ 2025-02-13).  Every function cites the reference file:line it follows.  The
 * restatement follows the reference's *portable* (scalar `#else`) bodies: the integer parts are
 * exactly those of the SIMD bodies the reference runs on x86; only the order of the final fp32 adds
 * differs (≈1e-7 relative).  quantize_row_* / dequantize_row_* are deterministic and are pinned
 * BIT-EXACT against the compiled reference (oracle/_ref, built by oracle/ref.mk) in
 * tests/test_oracle_vs_ref.py and against tests/golden/ fixtures generated from it.
 *
 * Formats: the five of the HIP path (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K) and the widening formats Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL, IQ4_XS.
 * Parity status: PINNED (bit-exact for quantize/dequantize rows, rel-L2 <= 2e-6 for mul_mat) against
 * the reference compiled here from /root/reference; the reference ships no stored golden vectors for
 * quantized MUL_MAT, so the fixtures under tests/golden/ were produced by running the
 * reference itself (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QK_K 256
#define K_SCALE_SIZE 12

/* ggml type ids — include/ggml.h:351-390 */
enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8, T_Q8_1 = 9, T_Q2_K = 10, T_Q3_K = 11,
       T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15, T_IQ4_NL = 20, T_IQ4_XS = 23 };

/* ---- block formats: src/ggml-common.h:161-328 ------------------------------------------------ */
#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; } block_q4_0;                                   /* :161-166 */
typedef struct { uint16_t d; int8_t qs[32]; } block_q8_0;                                    /* :203-208 */
typedef struct { uint16_t d, dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qs[QK_K / 2]; } block_q4_K; /* :279-290 */
typedef struct { uint16_t d, dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qh[QK_K / 8]; uint8_t qs[QK_K / 2]; } block_q5_K; /* :296-308 */
typedef struct { uint8_t ql[QK_K / 2]; uint8_t qh[QK_K / 4]; int8_t scales[QK_K / 16]; uint16_t d; } block_q6_K; /* :314-320 */
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } block_q8_K;            /* :323-328 */
/* the formats of SURVEY.md 8(f) rank 4 (oracle only so far: no HIP kernels take them yet) */
typedef struct { uint16_t d, m; uint8_t qs[16]; } block_q4_1;                                 /* :168-180 */
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; } block_q5_0;                     /* :182-188 */
typedef struct { uint16_t d, m; uint8_t qh[4]; uint8_t qs[16]; } block_q5_1;                  /* :190-202 */
typedef struct { uint16_t d, s; int8_t qs[32]; } block_q8_1;                                  /* :210-222 */
typedef struct { uint8_t scales[QK_K / 16]; uint8_t qs[QK_K / 4]; uint16_t d, dmin; } block_q2_K; /* :247-260 */
typedef struct { uint8_t hmask[QK_K / 8]; uint8_t qs[QK_K / 4]; uint8_t scales[12]; uint16_t d; } block_q3_K; /* :267-273 */
typedef struct { uint16_t d; uint8_t qs[16]; } block_iq4_nl;                                  /* :400-403 */
typedef struct { uint16_t d; uint16_t scales_h; uint8_t scales_l[QK_K / 64]; uint8_t qs[QK_K / 2]; } block_iq4_xs; /* :406-411 */
#pragma pack(pop)

/* ---- fp16 <-> fp32, IEEE round-to-nearest-even (semantics of GGML_FP32_TO_FP16 / _FP16_TO_FP32,
 *      src/ggml-impl.h:311-420: F16C _cvtss_sh / the bit-twiddling fallback are both RNE) -------- */
static inline float fp16_to_fp32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) { bits = sign; }
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) { bits = sign | 0x7F800000u | (man << 13); }
    else { bits = sign | ((exp + 112) << 23) | (man << 13); }
    float f; memcpy(&f, &bits, 4); return f;
}
static inline uint16_t fp32_to_fp16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00 | (x > 0x7F800000u ? 0x200 | ((x >> 13) & 0x3FF) : 0));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00);          /* rounds to >= 65520 -> inf */
    if (x < 0x33000001u) return sign;                                /* < 2^-25 (or == 2^-25: tie to even 0) */
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                   /* bits dropped */
    uint32_t half = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half & 1))) half++;
    if (e < -14) return (uint16_t)(sign | half);                     /* subnormal (may carry into normal) */
    return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (half - 0x400)));
}
float    oracle_fp16_to_fp32(uint16_t h) { return fp16_to_fp32(h); }
uint16_t oracle_fp32_to_fp16(float f)    { return fp32_to_fp16(f); }

/* src/ggml-quants.c:372-377 */
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

/* src/ggml-quants.c:631-638 */
static inline void get_scale_min_k4(int j, const uint8_t *q, uint8_t *d, uint8_t *m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

size_t oracle_type_size(int type) {      /* src/ggml.c type_traits table :568-... */
    switch (type) {
        case T_F32: return 4; case T_F16: return 2;
        case T_Q4_0: return sizeof(block_q4_0); case T_Q8_0: return sizeof(block_q8_0);
        case T_Q4_K: return sizeof(block_q4_K); case T_Q5_K: return sizeof(block_q5_K);
        case T_Q6_K: return sizeof(block_q6_K); case T_Q8_K: return sizeof(block_q8_K);
        case T_Q4_1: return sizeof(block_q4_1); case T_Q5_0: return sizeof(block_q5_0); case T_Q5_1: return sizeof(block_q5_1);
        case T_Q8_1: return sizeof(block_q8_1); case T_Q2_K: return sizeof(block_q2_K); case T_Q3_K: return sizeof(block_q3_K);
        case T_IQ4_NL: return sizeof(block_iq4_nl); case T_IQ4_XS: return sizeof(block_iq4_xs);
    }
    return 0;
}
int oracle_blck_size(int type) {
    switch (type) {
        case T_F32: case T_F16: return 1;
        case T_Q4_0: case T_Q8_0: case T_Q4_1: case T_Q5_0: case T_Q5_1: case T_Q8_1: case T_IQ4_NL: return 32;
        case T_Q4_K: case T_Q5_K: case T_Q6_K: case T_Q8_K: case T_Q2_K: case T_Q3_K: case T_IQ4_XS: return QK_K;
    }
    return 0;
}
size_t oracle_row_size(int type, int64_t k) { return (size_t)(k / oracle_blck_size(type)) * oracle_type_size(type); }

/* ============================ dequantize_row_* (normative layout) ============================== */
/* src/ggml-quants.c:255-273 */
void oracle_dequantize_row_q4_0(const void *vx, float *y, int64_t k) {
    const block_q4_0 *x = vx;
    for (int64_t i = 0; i < k / 32; i++) {
        const float d = fp16_to_fp32(x[i].d);
        for (int j = 0; j < 16; ++j) {
            const int x0 = (x[i].qs[j] & 0x0F) - 8, x1 = (x[i].qs[j] >> 4) - 8;
            y[i * 32 + j] = x0 * d; y[i * 32 + j + 16] = x1 * d;
        }
    }
}
/* src/ggml-quants.c:349-363 */
void oracle_dequantize_row_q8_0(const void *vx, float *y, int64_t k) {
    const block_q8_0 *x = vx;
    for (int64_t i = 0; i < k / 32; i++) {
        const float d = fp16_to_fp32(x[i].d);
        for (int j = 0; j < 32; ++j) y[i * 32 + j] = x[i].qs[j] * d;
    }
}
/* src/ggml-quants.c:1280-1302 */
void oracle_dequantize_row_q4_K(const void *vx, float *y, int64_t k) {
    const block_q4_K *x = vx;
    for (int64_t i = 0; i < k / QK_K; i++) {
        const uint8_t *q = x[i].qs;
        const float d = fp16_to_fp32(x[i].d), min = fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
            q += 32; is += 2;
        }
    }
}
/* src/ggml-quants.c:1482-1507 */
void oracle_dequantize_row_q5_K(const void *vx, float *y, int64_t k) {
    const block_q5_K *x = vx;
    for (int64_t i = 0; i < k / QK_K; i++) {
        const uint8_t *ql = x[i].qs, *qh = x[i].qh;
        const float d = fp16_to_fp32(x[i].d), min = fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m, u1 = 1, u2 = 2;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * ((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * ((ql[l] >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
            ql += 32; is += 2; u1 <<= 2; u2 <<= 2;
        }
    }
}
/* src/ggml-quants.c:1690-1719 */
void oracle_dequantize_row_q6_K(const void *vx, float *y, int64_t k) {
    const block_q6_K *x = vx;
    for (int64_t i = 0; i < k / QK_K; i++) {
        const float d = fp16_to_fp32(x[i].d);
        const uint8_t *ql = x[i].ql, *qh = x[i].qh; const int8_t *sc = x[i].scales;
        for (int n = 0; n < QK_K; n += 128) {
            for (int l = 0; l < 32; ++l) {
                int is = l / 16;
                const int8_t q1 = (int8_t)((ql[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t)((ql[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l + 0] = d * sc[is + 0] * q1; y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3; y[l + 96] = d * sc[is + 6] * q4;
            }
            y += 128; ql += 64; qh += 32; sc += 8;
        }
    }
}
/* src/ggml-quants.c:2518-2527 */
void oracle_dequantize_row_q8_K(const void *vx, float *y, int64_t k) {
    const block_q8_K *x = vx;
    for (int64_t i = 0; i < k / QK_K; i++)
        for (int j = 0; j < QK_K; ++j) *y++ = x[i].d * x[i].qs[j];
}
/* src/ggml-quants.c:275-293 */
void oracle_dequantize_row_q4_1(const void *vx, float *y, int64_t k) {
    const block_q4_1 *x = vx;
    for (int64_t i = 0; i < k / 32; i++) {
        const float d = fp16_to_fp32(x[i].d), m = fp16_to_fp32(x[i].m);
        for (int j = 0; j < 16; ++j) { y[i * 32 + j] = (x[i].qs[j] & 0x0F) * d + m; y[i * 32 + j + 16] = (x[i].qs[j] >> 4) * d + m; }
    }
}
/* the fifth bits of a 32-block: bit j of qh belongs to weight j (low half: j < 16 -> nibble j; high half: j + 16) */
static inline uint32_t load_qh(const uint8_t *qh) { uint32_t v; memcpy(&v, qh, 4); return v; }
/* src/ggml-quants.c:295-319 */
void oracle_dequantize_row_q5_0(const void *vx, float *y, int64_t k) {
    const block_q5_0 *x = vx;
    for (int64_t i = 0; i < k / 32; i++) {
        const float d = fp16_to_fp32(x[i].d); const uint32_t qh = load_qh(x[i].qh);
        for (int j = 0; j < 16; ++j) {
            const int x0 = ((x[i].qs[j] & 0x0F) | (((qh >> j) & 1) << 4)) - 16, x1 = ((x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16;
            y[i * 32 + j] = x0 * d; y[i * 32 + j + 16] = x1 * d;
        }
    }
}
/* src/ggml-quants.c:321-347 */
void oracle_dequantize_row_q5_1(const void *vx, float *y, int64_t k) {
    const block_q5_1 *x = vx;
    for (int64_t i = 0; i < k / 32; i++) {
        const float d = fp16_to_fp32(x[i].d), m = fp16_to_fp32(x[i].m); const uint32_t qh = load_qh(x[i].qh);
        for (int j = 0; j < 16; ++j) {
            const int x0 = (x[i].qs[j] & 0x0F) | (((qh >> j) & 1) << 4), x1 = (x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
            y[i * 32 + j] = x0 * d + m; y[i * 32 + j + 16] = x1 * d + m;
        }
    }
}
/* src/ggml-quants.c:712-744: sixteen 16-weight sub-blocks, scales[] = 4-bit scale | 4-bit min << 4; within each 128-half the
 * 2-bit fields of 32 bytes are walked shift 0,2,4,6, each shift covering two sub-blocks (bytes 0-15, 16-31) */
void oracle_dequantize_row_q2_K(const void *vx, float *y, int64_t k) {
    const block_q2_K *x = vx;
    for (int64_t i = 0; i < k / QK_K; i++) {
        const float d = fp16_to_fp32(x[i].d), dmin = fp16_to_fp32(x[i].dmin);
        for (int n = 0; n < 2; n++)
            for (int sh = 0; sh < 4; sh++)
                for (int half = 0; half < 2; half++) {
                    const uint8_t sc = x[i].scales[n * 8 + sh * 2 + half];
                    const float dl = d * (sc & 0xF), ml = dmin * (sc >> 4);
                    const uint8_t *q = x[i].qs + 32 * n + 16 * half;
                    for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l] >> (2 * sh)) & 3)) - ml;
                }
    }
}
/* the sixteen 6-bit scales of a Q3_K superblock: low 4 bits in scales[0..7] (two per byte), high 2 bits in scales[8..11]
 * (the aux/kmask shuffle of src/ggml-quants.c:1072-1077, written out per index) */
static inline int q3_scale(const uint8_t *sc, int j) {
    const int lo = j < 8 ? (sc[j] & 0xF) : (sc[j - 8] >> 4);
    const int hi = (sc[8 + (j & 3)] >> (2 * (j >> 2))) & 3;
    return (lo | (hi << 4)) - 32;
}
/* src/ggml-quants.c:1056-1104 */
void oracle_dequantize_row_q3_K(const void *vx, float *y, int64_t k) {
    const block_q3_K *x = vx;
    for (int64_t i = 0; i < k / QK_K; i++) {
        const float d_all = fp16_to_fp32(x[i].d);
        for (int n = 0; n < 2; n++)
            for (int sh = 0; sh < 4; sh++)
                for (int half = 0; half < 2; half++) {
                    const float dl = d_all * q3_scale(x[i].scales, n * 8 + sh * 2 + half);
                    const uint8_t *q = x[i].qs + 32 * n + 16 * half, *hm = x[i].hmask + 16 * half;
                    const uint8_t m = (uint8_t)(1u << (4 * n + sh));
                    for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l] >> (2 * sh)) & 3) - ((hm[l] & m) ? 0 : 4));
                }
    }
}
/* the non-linear 4-bit codebook, src/ggml-quants.c:2434 */
static const int8_t kvalues_iq4nl[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
/* src/ggml-quants.c:2436-2452 */
void oracle_dequantize_row_iq4_nl(const void *vx, float *y, int64_t k) {
    const block_iq4_nl *x = vx;
    for (int64_t i = 0; i < k / 32; i++) {
        const float d = fp16_to_fp32(x[i].d);
        for (int j = 0; j < 16; ++j) { y[i * 32 + j] = d * kvalues_iq4nl[x[i].qs[j] & 0xf]; y[i * 32 + j + 16] = d * kvalues_iq4nl[x[i].qs[j] >> 4]; }
    }
}
/* src/ggml-quants.c:2454-2475: eight 32-weight sub-blocks with 6-bit scales ls (low 4 bits in scales_l, high 2 in scales_h), dl = d * (ls - 32) */
void oracle_dequantize_row_iq4_xs(const void *vx, float *y, int64_t k) {
    const block_iq4_xs *x = vx;
    for (int64_t i = 0; i < k / QK_K; i++) {
        const float d = fp16_to_fp32(x[i].d);
        for (int ib = 0; ib < QK_K / 32; ++ib) {
            const int ls = ((x[i].scales_l[ib / 2] >> 4 * (ib % 2)) & 0xf) | (((x[i].scales_h >> 2 * ib) & 3) << 4);
            const float dl = d * (ls - 32);
            const uint8_t *qs = x[i].qs + 16 * ib;
            for (int j = 0; j < 16; ++j) { y[i * QK_K + 32 * ib + j] = dl * kvalues_iq4nl[qs[j] & 0xf]; y[i * QK_K + 32 * ib + j + 16] = dl * kvalues_iq4nl[qs[j] >> 4]; }
        }
    }
}
void oracle_dequantize_row(int type, const void *x, float *y, int64_t k) {
    switch (type) {
        case T_IQ4_NL: oracle_dequantize_row_iq4_nl(x, y, k); break; case T_IQ4_XS: oracle_dequantize_row_iq4_xs(x, y, k); break;
        case T_Q4_0: oracle_dequantize_row_q4_0(x, y, k); break; case T_Q8_0: oracle_dequantize_row_q8_0(x, y, k); break;
        case T_Q4_K: oracle_dequantize_row_q4_K(x, y, k); break; case T_Q5_K: oracle_dequantize_row_q5_K(x, y, k); break;
        case T_Q6_K: oracle_dequantize_row_q6_K(x, y, k); break; case T_Q8_K: oracle_dequantize_row_q8_K(x, y, k); break;
        case T_Q4_1: oracle_dequantize_row_q4_1(x, y, k); break; case T_Q5_0: oracle_dequantize_row_q5_0(x, y, k); break;
        case T_Q5_1: oracle_dequantize_row_q5_1(x, y, k); break; case T_Q2_K: oracle_dequantize_row_q2_K(x, y, k); break;
        case T_Q3_K: oracle_dequantize_row_q3_K(x, y, k); break;
        case T_F32: memcpy(y, x, (size_t)k * 4); break;
        case T_F16: for (int64_t i = 0; i < k; i++) y[i] = fp16_to_fp32(((const uint16_t *)x)[i]); break;
    }
}

/* ============================ quantize_row_* ================================================== */
/* src/ggml-quants.c:31-66 */
void oracle_quantize_row_q4_0_ref(const float *x, void *vy, int64_t k) {
    block_q4_0 *y = vy;
    for (int64_t i = 0; i < k / 32; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = x[i * 32 + j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
        const float d = max / -8; const float id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        for (int j = 0; j < 16; ++j) {
            const float x0 = x[i * 32 + 0 + j] * id, x1 = x[i * 32 + 16 + j] * id;
            const uint8_t xi0 = MIN(15, (int8_t)(x0 + 8.5f)), xi1 = MIN(15, (int8_t)(x1 + 8.5f));
            y[i].qs[j] = xi0 | (xi1 << 4);
        }
    }
}
/* src/ggml-quants.c:194-217 — the `_ref` (roundf, id = 1/d) variant */
/* src/ggml-quants.c:68-101 */
void oracle_quantize_row_q4_1_ref(const float *x, void *vy, int64_t k) {
    block_q4_1 *y = vy;
    for (int64_t i = 0; i < k / 32; i++) {
        float min = 3.402823466e+38f, max = -3.402823466e+38f;
        for (int j = 0; j < 32; j++) { const float v = x[i * 32 + j]; if (v < min) min = v; if (v > max) max = v; }
        const float d = (max - min) / ((1 << 4) - 1), id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d); y[i].m = fp32_to_fp16(min);
        for (int j = 0; j < 16; ++j) {
            const float x0 = (x[i * 32 + j] - min) * id, x1 = (x[i * 32 + 16 + j] - min) * id;
            const uint8_t xi0 = MIN(15, (int8_t)(x0 + 0.5f)), xi1 = MIN(15, (int8_t)(x1 + 0.5f));
            y[i].qs[j] = xi0 | (xi1 << 4);
        }
    }
}
/* src/ggml-quants.c:103-145 */
void oracle_quantize_row_q5_0_ref(const float *x, void *vy, int64_t k) {
    block_q5_0 *y = vy;
    for (int64_t i = 0; i < k / 32; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = x[i * 32 + j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
        const float d = max / -16, id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        uint32_t qh = 0;
        for (int j = 0; j < 16; ++j) {
            const float x0 = x[i * 32 + j] * id, x1 = x[i * 32 + 16 + j] * id;
            const uint8_t xi0 = MIN(31, (int8_t)(x0 + 16.5f)), xi1 = MIN(31, (int8_t)(x1 + 16.5f));
            y[i].qs[j] = (xi0 & 0x0F) | ((xi1 & 0x0F) << 4);
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0); qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}
/* src/ggml-quants.c:147-192 */
void oracle_quantize_row_q5_1_ref(const float *x, void *vy, int64_t k) {
    block_q5_1 *y = vy;
    for (int64_t i = 0; i < k / 32; i++) {
        float min = 3.402823466e+38f, max = -3.402823466e+38f;
        for (int j = 0; j < 32; j++) { const float v = x[i * 32 + j]; if (v < min) min = v; if (v > max) max = v; }
        const float d = (max - min) / ((1 << 5) - 1), id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d); y[i].m = fp32_to_fp16(min);
        uint32_t qh = 0;
        for (int j = 0; j < 16; ++j) {
            const float x0 = (x[i * 32 + j] - min) * id, x1 = (x[i * 32 + 16 + j] - min) * id;
            const uint8_t xi0 = (uint8_t)(x0 + 0.5f), xi1 = (uint8_t)(x1 + 0.5f);
            y[i].qs[j] = (xi0 & 0x0F) | ((xi1 & 0x0F) << 4);
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0); qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}
void oracle_quantize_row_q8_0_ref(const float *x, void *vy, int64_t k) {
    block_q8_0 *y = vy;
    for (int64_t i = 0; i < k / 32; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) amax = MAX(amax, fabsf(x[i * 32 + j]));
        const float d = amax / ((1 << 7) - 1); const float id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        for (int j = 0; j < 32; ++j) y[i].qs[j] = (int8_t)roundf(x[i * 32 + j] * id);
    }
}
/* src/ggml-cpu/ggml-cpu-quants.c:778-815 — the AVX2 body the CPU *backend* runs for MUL_MAT
 * activations: d = max/127 stored fp16, id = 127/max, round-to-nearest-even. */
void oracle_quantize_row_q8_0_cpu(const float *x, void *vy, int64_t k) {
    block_q8_0 *y = vy;
    for (int64_t i = 0; i < k / 32; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) amax = MAX(amax, fabsf(x[i * 32 + j]));
        const float d = amax / 127.f;
        y[i].d = fp32_to_fp16(d);
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        for (int j = 0; j < 32; ++j) y[i].qs[j] = (int8_t)(int)nearbyintf(x[i * 32 + j] * id);  /* RNE */
    }
}
/* src/ggml-cpu/ggml-cpu-quants.c:1076-1119 — the AVX2 body the CPU backend runs for the activations of Q4_1 / Q5_1 weights:
 * as q8_0_cpu, plus s = fp16(d * sum of the quants) with d still in fp32 */
void oracle_quantize_row_q8_1_cpu(const float *x, void *vy, int64_t k) {
    block_q8_1 *y = vy;
    for (int64_t i = 0; i < k / 32; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) amax = MAX(amax, fabsf(x[i * 32 + j]));
        const float d = amax / 127.f;
        y[i].d = fp32_to_fp16(d);
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        int sum = 0;
        for (int j = 0; j < 32; ++j) { const int q = (int)nearbyintf(x[i * 32 + j] * id); y[i].qs[j] = (int8_t)q; sum += q; }
        y[i].s = fp32_to_fp16(d * (float)sum);
    }
}
/* src/ggml-quants.c:2479-2516 (quantize_row_q8_K == _ref, src/ggml-cpu/ggml-cpu-quants.c:1646-1648) */
void oracle_quantize_row_q8_K(const float *x, void *vy, int64_t k) {
    block_q8_K *y = vy;
    for (int64_t i = 0; i < k / QK_K; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < QK_K; ++j) { float ax = fabsf(x[j]); if (ax > amax) { amax = ax; max = x[j]; } }
        if (!amax) { y[i].d = 0; memset(y[i].qs, 0, QK_K); memset(y[i].bsums, 0, sizeof(y[i].bsums)); x += QK_K; continue; }
        const float iscale = -127.f / max;
        for (int j = 0; j < QK_K; ++j) { int v = nearest_int(iscale * x[j]); y[i].qs[j] = MIN(127, v); }
        for (int j = 0; j < QK_K / 16; ++j) {
            int sum = 0; for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = sum;
        }
        y[i].d = 1 / iscale;
        x += QK_K;
    }
}

/* ============================ vec_dot (portable bodies) ======================================= */
/* src/ggml-cpu/ggml-cpu-quants.c:2293-2310 */
float oracle_vec_dot_q4_0_q8_0(int n, const void *vx, const void *vy) {
    const block_q4_0 *x = vx; const block_q8_0 *y = vy; float sumf = 0;
    for (int ib = 0; ib < n / 32; ++ib) {
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const int v0 = (x[ib].qs[j] & 0x0F) - 8, v1 = (x[ib].qs[j] >> 4) - 8;
            sumi0 += v0 * y[ib].qs[j]; sumi1 += v1 * y[ib].qs[j + 16];
        }
        sumf += (sumi0 + sumi1) * fp16_to_fp32(x[ib].d) * fp16_to_fp32(y[ib].d);
    }
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:3335-... scalar tail */
float oracle_vec_dot_q8_0_q8_0(int n, const void *vx, const void *vy) {
    const block_q8_0 *x = vx; const block_q8_0 *y = vy; float sumf = 0;
    for (int ib = 0; ib < n / 32; ++ib) {
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += x[ib].qs[j] * y[ib].qs[j];
        sumf += sumi * (fp16_to_fp32(x[ib].d) * fp16_to_fp32(y[ib].d));
    }
    return sumf;
}
static void unpack_scales_k4(const uint8_t *scales12, uint8_t *sc8, uint8_t *m8) {   /* the utmp/kmask shuffle, :6160-6165 */
    for (int j = 0; j < 8; j++) get_scale_min_k4(j, scales12, &sc8[j], &m8[j]);
}
/* src/ggml-cpu/ggml-cpu-quants.c:6137-6193 */
float oracle_vec_dot_q4_K_q8_K(int n, const void *vx, const void *vy) {
    const block_q4_K *x = vx; const block_q8_K *y = vy;
    float sums[8] = {0}; float sumf = 0;
    for (int i = 0; i < n / QK_K; ++i) {
        int8_t aux8[QK_K]; int32_t aux32[8] = {0}; uint8_t scales[8], mins[8];
        const uint8_t *q4 = x[i].qs; const int8_t *q8 = y[i].qs; int8_t *a = aux8;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] & 0xF); a += 32;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] >> 4); a += 32; q4 += 32;
        }
        unpack_scales_k4(x[i].scales, scales, mins);
        int sumi = 0;
        for (int j = 0; j < QK_K / 16; ++j) sumi += y[i].bsums[j] * mins[j / 2];
        a = aux8;
        for (int j = 0; j < QK_K / 32; ++j) {
            int32_t scale = scales[j];
            for (int r = 0; r < 4; r++) { for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t)(q8[l] * a[l]); q8 += 8; a += 8; }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:6769-6830 */
float oracle_vec_dot_q5_K_q8_K(int n, const void *vx, const void *vy) {
    const block_q5_K *x = vx; const block_q8_K *y = vy;
    float sums[8] = {0}; float sumf = 0;
    for (int i = 0; i < n / QK_K; ++i) {
        int8_t aux8[QK_K]; int32_t aux32[8] = {0}; uint8_t scales[8], mins[8];
        const uint8_t *q4 = x[i].qs, *hm = x[i].qh; const int8_t *q8 = y[i].qs; int8_t *a = aux8; uint8_t m = 1;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] & 0xF) + (hm[l] & m ? 16 : 0); a += 32; m <<= 1;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] >> 4) + (hm[l] & m ? 16 : 0); a += 32; m <<= 1;
            q4 += 32;
        }
        unpack_scales_k4(x[i].scales, scales, mins);
        int sumi = 0;
        for (int j = 0; j < QK_K / 16; ++j) sumi += y[i].bsums[j] * mins[j / 2];
        a = aux8;
        for (int j = 0; j < QK_K / 32; ++j) {
            int32_t scale = scales[j];
            for (int r = 0; r < 4; r++) { for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t)(q8[l] * a[l]); q8 += 8; a += 8; }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:7425-7467 */
float oracle_vec_dot_q6_K_q8_K(int n, const void *vx, const void *vy) {
    const block_q6_K *x = vx; const block_q8_K *y = vy;
    float sums[8] = {0}; float sumf = 0;
    for (int i = 0; i < n / QK_K; ++i) {
        int8_t aux8[QK_K]; int32_t aux32[8] = {0};
        const uint8_t *q4 = x[i].ql, *qh = x[i].qh; const int8_t *q8 = y[i].qs; int8_t *a = aux8;
        for (int j = 0; j < QK_K; j += 128) {
            for (int l = 0; l < 32; ++l) {
                a[l + 0] = (int8_t)((q4[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                a[l + 32] = (int8_t)((q4[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                a[l + 64] = (int8_t)((q4[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                a[l + 96] = (int8_t)((q4[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            }
            a += 128; q4 += 64; qh += 32;
        }
        a = aux8;
        for (int j = 0; j < QK_K / 16; ++j) {
            int scale = x[i].scales[j];
            for (int r = 0; r < 2; r++) { for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t)(q8[l] * a[l]); q8 += 8; a += 8; }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

/* src/ggml-cpu/ggml-cpu-quants.c:2585-2601 */
float oracle_vec_dot_q4_1_q8_1(int n, const void *vx, const void *vy) {
    const block_q4_1 *x = vx; const block_q8_1 *y = vy; float sumf = 0;
    for (int ib = 0; ib < n / 32; ++ib) {
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) { sumi0 += (x[ib].qs[j] & 0x0F) * y[ib].qs[j]; sumi1 += (x[ib].qs[j] >> 4) * y[ib].qs[j + 16]; }
        sumf += (fp16_to_fp32(x[ib].d) * fp16_to_fp32(y[ib].d)) * (sumi0 + sumi1) + fp16_to_fp32(x[ib].m) * fp16_to_fp32(y[ib].s);
    }
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:2935-2957 */
float oracle_vec_dot_q5_0_q8_0(int n, const void *vx, const void *vy) {
    const block_q5_0 *x = vx; const block_q8_0 *y = vy; float sumf = 0;
    for (int ib = 0; ib < n / 32; ++ib) {
        const uint32_t qh = load_qh(x[ib].qh);
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const int x0 = ((x[ib].qs[j] & 0x0F) | (((qh >> j) & 1) << 4)) - 16, x1 = ((x[ib].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16;
            sumi0 += x0 * y[ib].qs[j]; sumi1 += x1 * y[ib].qs[j + 16];
        }
        sumf += (fp16_to_fp32(x[ib].d) * fp16_to_fp32(y[ib].d)) * (sumi0 + sumi1);
    }
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:3309-3331 */
float oracle_vec_dot_q5_1_q8_1(int n, const void *vx, const void *vy) {
    const block_q5_1 *x = vx; const block_q8_1 *y = vy; float sumf = 0;
    for (int ib = 0; ib < n / 32; ++ib) {
        const uint32_t qh = load_qh(x[ib].qh);
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const int x0 = (x[ib].qs[j] & 0x0F) | (((qh >> j) & 1) << 4), x1 = (x[ib].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
            sumi0 += x0 * y[ib].qs[j]; sumi1 += x1 * y[ib].qs[j + 16];
        }
        sumf += (fp16_to_fp32(x[ib].d) * fp16_to_fp32(y[ib].d)) * (sumi0 + sumi1) + fp16_to_fp32(x[ib].m) * fp16_to_fp32(y[ib].s);
    }
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:4725-4764 */
float oracle_vec_dot_q2_K_q8_K(int n, const void *vx, const void *vy) {
    const block_q2_K *x = vx; const block_q8_K *y = vy; float sumf = 0;
    for (int i = 0; i < n / QK_K; ++i) {
        const uint8_t *sc = x[i].scales; const int8_t *q8 = y[i].qs;
        int summs = 0;
        for (int j = 0; j < 16; ++j) summs += y[i].bsums[j] * (sc[j] >> 4);
        const float dall = y[i].d * fp16_to_fp32(x[i].d), dmin = y[i].d * fp16_to_fp32(x[i].dmin);
        int isum = 0;
        for (int nn = 0; nn < 2; nn++)
            for (int sh = 0; sh < 4; sh++)
                for (int half = 0; half < 2; half++) {
                    const uint8_t *q2 = x[i].qs + 32 * nn + 16 * half;
                    int isuml = 0;
                    for (int l = 0; l < 16; ++l) isuml += q8[l] * ((q2[l] >> (2 * sh)) & 3);
                    isum += (sc[nn * 8 + sh * 2 + half] & 0xF) * isuml;
                    q8 += 16;
                }
        sumf += dall * isum - dmin * summs;
    }
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:5482-5545: integer sums in eight lanes (lane = position & 7), scaled per superblock into eight
 * float partial sums that are added at the very end */
float oracle_vec_dot_q3_K_q8_K(int n, const void *vx, const void *vy) {
    const block_q3_K *x = vx; const block_q8_K *y = vy;
    float sums[8] = {0};
    for (int i = 0; i < n / QK_K; ++i) {
        int32_t aux32[8] = {0};
        const int8_t *q8 = y[i].qs;
        for (int nn = 0; nn < 2; nn++)
            for (int sh = 0; sh < 4; sh++)
                for (int half = 0; half < 2; half++) {
                    const int scale = q3_scale(x[i].scales, nn * 8 + sh * 2 + half);
                    const uint8_t *q3 = x[i].qs + 32 * nn + 16 * half, *hm = x[i].hmask + 16 * half;
                    const uint8_t m = (uint8_t)(1u << (4 * nn + sh));
                    for (int l = 0; l < 16; ++l) {
                        const int a = (int)((q3[l] >> (2 * sh)) & 3) - ((hm[l] & m) ? 0 : 4);
                        aux32[l & 7] += scale * (int16_t)(q8[l] * a);
                    }
                    q8 += 16;
                }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    float sumf = 0;
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

/* src/ggml-cpu/ggml-cpu-quants.c:10370-10561 (scalar tail :10550-10559) */
float oracle_vec_dot_iq4_nl_q8_0(int n, const void *vx, const void *vy) {
    const block_iq4_nl *x = vx; const block_q8_0 *y = vy; float sumf = 0;
    for (int ib = 0; ib < n / 32; ++ib) {
        const float d = fp16_to_fp32(y[ib].d) * fp16_to_fp32(x[ib].d);
        int sumi1 = 0, sumi2 = 0;
        for (int j = 0; j < 16; ++j) { sumi1 += y[ib].qs[j] * kvalues_iq4nl[x[ib].qs[j] & 0xf]; sumi2 += y[ib].qs[j + 16] * kvalues_iq4nl[x[ib].qs[j] >> 4]; }
        sumf += d * (sumi1 + sumi2);
    }
    return sumf;
}
/* src/ggml-cpu/ggml-cpu-quants.c:10563-10898 (scalar body :10866-10896): per superblock d4d8 = d * y.d, per 32-sub-block the two 16-halves
 * accumulated separately, sumf += d4d8 * (ls - 32) * (sumi1 + sumi2) */
float oracle_vec_dot_iq4_xs_q8_K(int n, const void *vx, const void *vy) {
    const block_iq4_xs *x = vx; const block_q8_K *y = vy; float sumf = 0;
    for (int ibl = 0; ibl < n / QK_K; ++ibl) {
        const float d4d8 = fp16_to_fp32(x[ibl].d) * y[ibl].d;
        uint16_t h = x[ibl].scales_h;
        const uint8_t *qs = x[ibl].qs; const int8_t *q8 = y[ibl].qs;
        for (int ib = 0; ib < QK_K / 32; ib += 2) {
            const uint8_t ls1 = (x[ibl].scales_l[ib / 2] & 0xf) | ((h << 4) & 0x30);
            const uint8_t ls2 = (x[ibl].scales_l[ib / 2] >> 4) | ((h << 2) & 0x30);
            h >>= 4;
            const float d1 = d4d8 * (ls1 - 32), d2 = d4d8 * (ls2 - 32);
            int sumi1 = 0, sumi2 = 0;
            for (int j = 0; j < 16; ++j) { sumi1 += q8[j] * kvalues_iq4nl[qs[j] & 0xf]; sumi2 += q8[j + 16] * kvalues_iq4nl[qs[j] >> 4]; }
            sumf += d1 * (sumi1 + sumi2);
            qs += 16; q8 += 32;
            sumi1 = sumi2 = 0;
            for (int j = 0; j < 16; ++j) { sumi1 += q8[j] * kvalues_iq4nl[qs[j] & 0xf]; sumi2 += q8[j + 16] * kvalues_iq4nl[qs[j] >> 4]; }
            sumf += d2 * (sumi1 + sumi2);
            qs += 16; q8 += 32;
        }
    }
    return sumf;
}

/* type_traits_cpu[]: weight type -> (vec_dot, vec_dot_type) — src/ggml-cpu/ggml-cpu.c:253-418 */
int oracle_vec_dot_type(int type) {
    switch (type) {
        case T_Q4_0: case T_Q8_0: case T_Q5_0: case T_IQ4_NL: return T_Q8_0; case T_Q4_1: case T_Q5_1: return T_Q8_1;
        case T_Q4_K: case T_Q5_K: case T_Q6_K: case T_Q2_K: case T_Q3_K: case T_IQ4_XS: return T_Q8_K;
    }
    return -1;
}
void oracle_quantize_act(int wtype, const float *x, void *y, int64_t k) {
    const int at = oracle_vec_dot_type(wtype);
    if (at == T_Q8_0) oracle_quantize_row_q8_0_cpu(x, y, k); else if (at == T_Q8_1) oracle_quantize_row_q8_1_cpu(x, y, k); else oracle_quantize_row_q8_K(x, y, k);
}
float oracle_vec_dot(int wtype, int n, const void *vx, const void *vy) {
    switch (wtype) {
        case T_Q4_0: return oracle_vec_dot_q4_0_q8_0(n, vx, vy); case T_Q8_0: return oracle_vec_dot_q8_0_q8_0(n, vx, vy);
        case T_Q4_K: return oracle_vec_dot_q4_K_q8_K(n, vx, vy); case T_Q5_K: return oracle_vec_dot_q5_K_q8_K(n, vx, vy);
        case T_Q6_K: return oracle_vec_dot_q6_K_q8_K(n, vx, vy);
        case T_Q4_1: return oracle_vec_dot_q4_1_q8_1(n, vx, vy); case T_Q5_0: return oracle_vec_dot_q5_0_q8_0(n, vx, vy);
        case T_Q5_1: return oracle_vec_dot_q5_1_q8_1(n, vx, vy); case T_Q2_K: return oracle_vec_dot_q2_K_q8_K(n, vx, vy);
        case T_Q3_K: return oracle_vec_dot_q3_K_q8_K(n, vx, vy);
        case T_IQ4_NL: return oracle_vec_dot_iq4_nl_q8_0(n, vx, vy); case T_IQ4_XS: return oracle_vec_dot_iq4_xs_q8_K(n, vx, vy);
    }
    return NAN;
}

/* ============================ MUL_MAT ======================================================== */
/* ggml_compute_forward_mul_mat, src/ggml-cpu/ggml-cpu.c:7428-7605: quantize every src1 row to
 * vec_dot_type (:7490-7509), then dst[col*M + row] = vec_dot(W row, quantized X col) (:7417).
 * W: M rows of K weights (row stride = oracle_row_size), X: B rows of K floats, Y: [B][M] floats. */
void oracle_mul_mat(int wtype, const void *W, const float *X, float *Y, int64_t M, int64_t K, int64_t B) {
    const int at = oracle_vec_dot_type(wtype);
    const size_t wrow = oracle_row_size(wtype, K), arow = oracle_row_size(at, K);
    uint8_t *act = malloc(arow * (size_t)B);
    for (int64_t b = 0; b < B; b++) oracle_quantize_act(wtype, X + b * K, act + b * arow, K);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++)
        for (int64_t m = 0; m < M; m++)
            Y[b * M + m] = oracle_vec_dot(wtype, (int)K, (const uint8_t *)W + m * wrow, act + b * arow);
    free(act);
}
/* second oracle ("exact"): dequantize_row + fp64 dot — SURVEY.md §8c caveat (ii) */
void oracle_mul_mat_exact(int wtype, const void *W, const float *X, float *Y, int64_t M, int64_t K, int64_t B) {
    const size_t wrow = oracle_row_size(wtype, K);
#pragma omp parallel
    {
        float *wf = malloc((size_t)K * 4);
#pragma omp for schedule(static)
        for (int64_t m = 0; m < M; m++) {
            oracle_dequantize_row(wtype, (const uint8_t *)W + m * wrow, wf, K);
            for (int64_t b = 0; b < B; b++) {
                double s = 0; for (int64_t k = 0; k < K; k++) s += (double)wf[k] * (double)X[b * K + k];
                Y[b * M + m] = (float)s;
            }
        }
        free(wf);
    }
}
/* ggml_compute_forward_mul_mat_id, src/ggml-cpu/ggml-cpu.c:7609-7784.
 * as: [n_expert][M rows of K]; b: [n_tok][n_b][K] f32 (n_b == n_used or 1, broadcast :7752);
 * ids: [n_tok][n_used] i32; dst: [n_tok][n_used][M]. */
void oracle_mul_mat_id(int wtype, const void *as, const float *b, const int32_t *ids, float *dst,
                       int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok) {
    const int at = oracle_vec_dot_type(wtype);
    const size_t wrow = oracle_row_size(wtype, K), arow = oracle_row_size(at, K);
    (void)n_expert;
    uint8_t *act = malloc(arow);
    for (int64_t t = 0; t < n_tok; t++)
        for (int64_t u = 0; u < n_used; u++) {
            const int32_t e = ids[t * n_used + u];
            const float *xcol = b + (t * n_b + (u % n_b)) * K;
            oracle_quantize_act(wtype, xcol, act, K);
            const uint8_t *We = (const uint8_t *)as + (size_t)e * M * wrow;
            for (int64_t m = 0; m < M; m++) dst[(t * n_used + u) * M + m] = oracle_vec_dot(wtype, (int)K, We + m * wrow, act);
        }
    free(act);
}

/* ============================ FLASH_ATTN_EXT ============================ */
/* ggml_compute_forward_flash_attn_ext_f16, src/ggml-cpu/ggml-cpu.c:10805-11016, for F16 K and V (contiguous tensors):
 *   q    [n_batch][n_head][n_q][D]        f32      (ggml ne = D, n_q, n_head, n_batch)
 *   k, v [n_batch_kv][n_head_kv][n_kv][D] fp16 bits
 *   mask [>= n_q][n_kv]                   fp16 bits or NULL
 *   dst  [n_batch][n_q][n_head][D]        f32      (the permute(0, 2, 1, 3) of :11012)
 * Per query row: Q -> fp16 (:10929, from_float of the F16 vec_dot_type), s = ggml_vec_dot_f16(k, Q) (src/ggml-cpu/ggml-cpu.c:1457-1497: fp32 products,
 * here summed in double like its scalar tail) * scale [softcap * tanhf] + slope * mask, -inf mask entries skipped (:10935-10938), online
 * softmax with the O accumulator kept in FP16 (:10960-10974: ggml_vec_scale_f16 / ggml_vec_mad_f16 compute in fp32 per element — fused
 * multiply-add on the x86 F16C paths, ggml-cpu.c:1585-1614,1700-1727 with GGML_F16_VEC_FMA = _mm512/_mm256_fmadd_ps :619,664 — and round
 * back to fp16), V /= S (:11001-11003). */
void oracle_flash_attn_ext_f16(const float *q, const uint16_t *k, const uint16_t *v, const uint16_t *mask, float *dst,
                               int64_t D, int64_t n_q, int64_t n_head, int64_t n_batch, int64_t n_kv, int64_t n_head_kv, int64_t n_batch_kv,
                               float scale, float max_bias, float logit_softcap) {
    if (logit_softcap != 0) scale /= logit_softcap;                                              /* :10876-10878 */
    const uint32_t n_head_log2 = 1u << (uint32_t)floor(log2((double)n_head));                    /* :10880-10884 */
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    const int64_t rk2 = n_head / n_head_kv, rk3 = n_batch / n_batch_kv;
    uint16_t *Qh = malloc((size_t)D * 2), *acc = malloc((size_t)D * 2);
    for (int64_t i3 = 0; i3 < n_batch; i3++)
        for (int64_t i2 = 0; i2 < n_head; i2++)
            for (int64_t i1 = 0; i1 < n_q; i1++) {
                const uint32_t h = (uint32_t)i2;
                const float slope = (max_bias > 0.0f) ? (h < n_head_log2 ? powf(m0, h + 1) : powf(m1, 2 * (h - n_head_log2) + 1)) : 1.0f;   /* :10902 */
                const float *pq = q + ((i3 * n_head + i2) * n_q + i1) * D;
                for (int64_t d = 0; d < D; d++) { Qh[d] = fp32_to_fp16(pq[d]); acc[d] = 0; }
                const uint16_t *kh = k + ((i3 / rk3) * n_head_kv + (i2 / rk2)) * n_kv * D, *vh = v + ((i3 / rk3) * n_head_kv + (i2 / rk2)) * n_kv * D;
                float S = 0.0f, M = -INFINITY;
                for (int64_t ic = 0; ic < n_kv; ic++) {
                    const float mv = mask ? slope * fp16_to_fp32(mask[i1 * n_kv + ic]) : 0.0f;
                    if (mv == -INFINITY) continue;
                    double sd = 0;
                    for (int64_t d = 0; d < D; d++) sd += (double)(fp16_to_fp32(kh[ic * D + d]) * fp16_to_fp32(Qh[d]));
                    float s = (float)sd * scale;
                    if (logit_softcap != 0.0f) s = logit_softcap * tanhf(s);
                    s += mv;
                    const float Mold = M;
                    float ms = 1.0f, vs = 1.0f;
                    if (s > M) {
                        M = s; ms = expf(Mold - M);
                        for (int64_t d = 0; d < D; d++) acc[d] = fp32_to_fp16(fp16_to_fp32(acc[d]) * ms);
                    } else vs = expf(s - M);
                    for (int64_t d = 0; d < D; d++) acc[d] = fp32_to_fp16(fmaf(fp16_to_fp32(vh[ic * D + d]), vs, fp16_to_fp32(acc[d])));
                    S = S * ms + vs;
                }
                const float S_inv = 1.0f / S;
                float *out = dst + ((i3 * n_q + i1) * n_head + i2) * D;
                for (int64_t d = 0; d < D; d++) out[d] = fp16_to_fp32(acc[d]) * S_inv;
            }
    free(Qh); free(acc);
}
