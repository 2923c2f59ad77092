// convert_w.hip — EXACT re-encodings of weight formats that have no MFMA prefill kernel of their own into one that has
// ("other formats reuse the GEMV/GEMM skeleton"):
//
//    Q5_0 -> Q8_0   w = d (q5 - 16)              = d q8,                 q8 = q5 - 16 in [-16, 15]         (same fp16 d)
//    Q3_K -> Q6_K   w = d (sc6 - 32) (q3 - 4)    = d sc8 (q6 - 32),      sc8 = sc6 - 32 in [-32, 31],
//                                                                         q6 = q3 + 28 in [28, 35]          (same fp16 d, same 16-weight groups)
//    Q2_K -> Q6_K + Q6_K   w = d sc4 q2 - dmin m4 = d sc4 (qa - 32)  +  dmin (-m4) (33 - 32)             (library-internal: see below)
//    IQ4_NL -> Q8_0 w = d kvalues[code]          = d q8,                 q8 = kvalues_iq4nl[code] in [-127, 113]  (same fp16 d)
//    Q4_1 / Q5_1 -> Q8_0 + Q8_0   w = d q + m    = d q8  +  m e0,        q8 = q in [0, 15] / [0, 31], e0 = [1, 0 x 31]   (library-internal: against the
//                                                                         activation image [x~ | s e0], s = the CPU's block_q8_1.s — see k_convert_q41_q8_0x2)
//    IQ4_XS -> Q6_K + Q6_K   w = d (ls - 32) kvalues[code] = d (4 (ls - 32)) (qa - 32) + d (ls - 32) (qb - 32),  kvalues = 4 (qa - 32) + (qb - 32)   (the same)
//
// Every weight keeps its value bit for bit (dequantize_row_q5_0 / _q3_K of the source == dequantize_row_q8_0 / _q6_K of the result,
// src/ggml-quants.c:307-331, 1139-1188 vs :349-363, 1690-1719), and source and target share their activation format on the CPU
// (vec_dot_type Q8_0 for Q5_0 / Q8_0, Q8_K for Q3_K / Q6_K: src/ggml-cpu/ggml-cpu.c:277-302, 318-341), so the prefill product through
// the target's GEMM is the product the reference defines for the source format.  The conversion runs per call into library scratch
// (one read of W + one write of the larger encoding, a few microseconds at 4096 x 4096) — against 64 passes of the 8-column GEMV
// over the weights at B = 512 without it.
//
// Q2_K has a second multiplier (dmin) that no single Q6_K superblock can carry, but the two TERMS of a Q2_K weight are each a Q6_K weight:
// the scale term with q6 = q2 + 32 and int8 scale sc4, and the minimum term — constant over a 16-weight group — with q6 = 33 ("1"), int8
// scale -m4 and dmin in the place of d.  dequantize_row_q2_K computes dl * q - ml with dl = d * sc4, ml = dmin * m4
// (src/ggml-quants.c:712-744); dequantize_row_q6_K of the two parts gives d * sc4 * q and dmin * (-m4) * 1 = -ml, and x + (-y) = x - y in
// IEEE arithmetic (value for value; only a ZERO result can carry the other sign).  So W.x = [A | B].[x ; x]: the re-encoded matrix has 2 K columns (row = the K / 256 scale superblocks followed by the
// K / 256 minimum superblocks) and multiplies an activation image in which x appears twice (capi.hip duplicates the fp16 image; it is
// panel-major in k, so the copy is one contiguous block).  Kept inside the library: the public ggml_cdna4_convert_weights offers only the
// same-shape re-encodings.
#include "cdna4_common.h"
#include "cdna4_kernels.h"

// one thread per 32-weight block: 22 bytes {fp16 d, qh[4], qs[16]} in, 34 bytes {fp16 d, int8 qs[32]} out (both 2-byte aligned)
__global__ __launch_bounds__(256) void k_convert_q5_0_q8_0(const uint8_t *__restrict__ W, int64_t w_row_bytes, int M, int nblk, uint8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * nblk) return;
    const int row = (int)(t / nblk), b = (int)(t % nblk);
    const uint8_t *src = W + (int64_t)row * w_row_bytes + (int64_t)b * 22;
    uint8_t *dst = out + ((int64_t)row * nblk + b) * 34;
    const uint32_t qh = ld_u32_a2(src + 2);
    uint32_t lo[4], hi[4];                                             // weights 4i..4i+3 / 16+4i..16+4i+3 as int8 bytes
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t q = ld_u32_a2(src + 6 + 4 * i);
        // h0 / h1: bit (4i + e) / (16 + 4i + e) of qh in bit 0 of byte e.  Byte-wise (nibble | h << 4) - 16 as int8:
        // h = 1: the nibble itself (0..15); h = 0: nibble - 16 = 0xF0 | nibble
        const uint32_t h0 = ((qh >> (4 * i)) & 0xFu) * 0x00204081u & 0x01010101u, h1 = ((qh >> (16 + 4 * i)) & 0xFu) * 0x00204081u & 0x01010101u;
        const uint32_t v0 = q & 0x0F0F0F0Fu, v1 = (q >> 4) & 0x0F0F0F0Fu;
        lo[i] = v0 | (((h0 ^ 0x01010101u) * 0xF0u));
        hi[i] = v1 | (((h1 ^ 0x01010101u) * 0xF0u));
    }
    *reinterpret_cast<uint16_t *>(dst) = ld_u16(src);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        *reinterpret_cast<uint16_t *>(dst + 2 + 4 * i) = (uint16_t)lo[i]; *reinterpret_cast<uint16_t *>(dst + 4 + 4 * i) = (uint16_t)(lo[i] >> 16);
        *reinterpret_cast<uint16_t *>(dst + 18 + 4 * i) = (uint16_t)hi[i]; *reinterpret_cast<uint16_t *>(dst + 20 + 4 * i) = (uint16_t)(hi[i] >> 16);
    }
}

// Q4_1 / Q5_1 (FIVE): one thread per 32-weight block {fp16 d, fp16 m, [qh[4],] qs[16]} -> block b of the row's scale part {d, int8 q} and block
// nblk + b of its minimum part {m, [1, 0 x 31]}: 2 K columns per row, against an activation image [x~ | s e0] whose second half holds, per 32-block, the
// CPU's own block_q8_1.s = fp16(d_x * sum q_x) in the block's first column and zeros elsewhere (k_quantize_q8_1, two_part).  So
//     W . x = sum_blocks [ sum_k (d q_k) x~_k  +  m * s ]
// — term for term what vec_dot_q4_1_q8_1 / q5_1_q8_1 add up (src/ggml-cpu/ggml-cpu-quants.c:2585-2601, 3309-3331), the m * s products exact in fp32.
// Why not [d q | m 1] against x twice (round 2), or a centred split (tried in round 3): for activations with a non-zero mean the CPU's result is
// dominated by ITS rounding of s to fp16 (|m| |s| 2^-12 per block); a product on the exact sum of x~ lands 1.4e-3 .. 2.9e-3 from the CPU backend
// (measured on MI355X in the reference's gpt-2 graph: MUL_MAT 1.4e-3 .. 1.5e-3 for Q4_1 / Q5_1 either way) — with the CPU's own s: 6e-4 .. 8e-4.
template <bool FIVE>
__global__ __launch_bounds__(256) void k_convert_q41_q8_0x2(const uint8_t *__restrict__ W, int64_t w_row_bytes, int M, int nblk, uint8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * nblk) return;
    const int row = (int)(t / nblk), b = (int)(t % nblk);
    const uint8_t *src = W + (int64_t)row * w_row_bytes + (int64_t)b * (FIVE ? 24 : 20);
    uint8_t *da = out + ((int64_t)row * 2 * nblk + b) * 34, *db = da + (int64_t)nblk * 34;
    const uint32_t qh = FIVE ? ld_u32_a2(src + 4) : 0u;
    *reinterpret_cast<uint16_t *>(da) = ld_u16(src);
    *reinterpret_cast<uint16_t *>(db) = ld_u16(src + 2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t q = ld_u32_a2(src + (FIVE ? 8 : 4) + 4 * i);
        const uint32_t h0 = ((qh >> (4 * i)) & 0xFu) * 0x00204081u & 0x01010101u, h1 = ((qh >> (16 + 4 * i)) & 0xFu) * 0x00204081u & 0x01010101u;
        const uint32_t lo = (q & 0x0F0F0F0Fu) | (h0 << 4), hi = ((q >> 4) & 0x0F0F0F0Fu) | (h1 << 4);      // weights 4i.. / 16+4i.. as bytes 0..15 / 0..31
        *reinterpret_cast<uint16_t *>(da + 2 + 4 * i) = (uint16_t)lo; *reinterpret_cast<uint16_t *>(da + 4 + 4 * i) = (uint16_t)(lo >> 16);
        *reinterpret_cast<uint16_t *>(da + 18 + 4 * i) = (uint16_t)hi; *reinterpret_cast<uint16_t *>(da + 20 + 4 * i) = (uint16_t)(hi >> 16);
    }
    *reinterpret_cast<uint16_t *>(db + 2) = (uint16_t)0x0001u;          // [1, 0, 0, ...]: m meets s, nothing else
#pragma unroll
    for (int i = 1; i < 16; i++) *reinterpret_cast<uint16_t *>(db + 2 + 2 * i) = (uint16_t)0u;
}

// IQ4_NL: one thread per 32-weight block {fp16 d, qs[16]} -> {d, int8 qs[32]} with q8 = kvalues_iq4nl[code]: w = d * kvalues[code]
// (dequantize_row_iq4_nl, src/ggml-quants.c:2436-2452) = d * q8 (dequantize_row_q8_0), same fp16 d, same fp32 product
__global__ __launch_bounds__(256) void k_convert_iq4_nl_q8_0(const uint8_t *__restrict__ W, int64_t w_row_bytes, int M, int nblk, uint8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * nblk) return;
    const int row = (int)(t / nblk), b = (int)(t % nblk);
    const uint8_t *src = W + (int64_t)row * w_row_bytes + (int64_t)b * 18;
    uint8_t *dst = out + ((int64_t)row * nblk + b) * 34;
    *reinterpret_cast<uint16_t *>(dst) = ld_u16(src);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t q = ld_u32_a2(src + 2 + 4 * i);
        const uint32_t lo = iq4nl_lut4(q & 0x0F0F0F0Fu), hi = iq4nl_lut4((q >> 4) & 0x0F0F0F0Fu);
        *reinterpret_cast<uint16_t *>(dst + 2 + 4 * i) = (uint16_t)lo; *reinterpret_cast<uint16_t *>(dst + 4 + 4 * i) = (uint16_t)(lo >> 16);
        *reinterpret_cast<uint16_t *>(dst + 18 + 4 * i) = (uint16_t)hi; *reinterpret_cast<uint16_t *>(dst + 20 + 4 * i) = (uint16_t)(hi >> 16);
    }
}

// 18 threads per superblock: 110 bytes {hmask[32], qs[64], scales[12], fp16 d} in, 210 bytes {ql[128], qh[64], int8 scales[16], fp16 d} out.
// Threads 0..15: (128-half n = t >> 3, bytes l = 4 (t & 7) .. + 3): weight 128 n + 32 j + l is Q3_K's (qs[32 n + l] >> 2 j) & 3 with hmask bit
// 4 n + j of hmask[l], and Q6_K's q_(j+1) of the same (n, l): low nibbles in ql[64 n + 32 (j & 1) + l] (high half of the byte for j >= 2),
// bits 4-5 in qh[32 n + l] >> 2 j.   Thread 16: the sixteen scales.   Thread 17: d.
__global__ __launch_bounds__(256) void k_convert_q3_K_q6_K(const uint8_t *__restrict__ W, int64_t w_row_bytes, int M, int nsb, uint8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * nsb * 18) return;
    const int pc = (int)(t % 18); const int64_t u = t / 18;
    const int row = (int)(u / nsb), sb = (int)(u % nsb);
    const uint8_t *src = W + (int64_t)row * w_row_bytes + (int64_t)sb * 110;
    uint8_t *dst = out + ((int64_t)row * nsb + sb) * 210;
    auto st32 = [](uint8_t *p, uint32_t v) { *reinterpret_cast<uint16_t *>(p) = (uint16_t)v; *reinterpret_cast<uint16_t *>(p + 2) = (uint16_t)(v >> 16); };
    if (pc < 16) {
        const int n = pc >> 3, l = 4 * (pc & 7);
        const uint32_t q = ld_u32_a2(src + 32 + 32 * n + l), hm = ld_u32_a2(src + l);
        uint32_t v[4];                                                  // q6 = (2 bits | hmask bit << 2) + 28, four bytes at a time
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = (((q >> (2 * j)) & 0x03030303u) | (((hm >> (4 * n + j)) & 0x01010101u) << 2)) + 0x1C1C1C1Cu;
        st32(dst + 64 * n + l, (v[0] & 0x0F0F0F0Fu) | ((v[2] & 0x0F0F0F0Fu) << 4));
        st32(dst + 64 * n + 32 + l, (v[1] & 0x0F0F0F0Fu) | ((v[3] & 0x0F0F0F0Fu) << 4));
        st32(dst + 128 + 32 * n + l, ((v[0] >> 4) & 0x03030303u) | (((v[1] >> 4) & 0x03030303u) << 2) | (((v[2] >> 4) & 0x03030303u) << 4) | (((v[3] >> 4) & 0x03030303u) << 6));
    } else if (pc == 16) {
        // scale i: low nibble = scales[i] & 0xF (i < 8) or scales[i - 8] >> 4; bits 4-5 = (scales[8 + i % 4] >> 2 (i / 4)) & 3; stored - 32
        const uint32_t a = ld_u32_a2(src + 96), b = ld_u32_a2(src + 100), c = ld_u32_a2(src + 104);
#pragma unroll
        for (int g = 0; g < 4; g++) {                                   // scales 4 g .. 4 g + 3
            const uint32_t lo = g == 0 ? (a & 0x0F0F0F0Fu) : (g == 1 ? (b & 0x0F0F0F0Fu) : (g == 2 ? ((a >> 4) & 0x0F0F0F0Fu) : ((b >> 4) & 0x0F0F0F0Fu)));
            const uint32_t s6 = lo | (((c >> (2 * g)) & 0x03030303u) << 4);
            // byte-wise s6 - 32 (0..63 -> -32..31): s6 >= 32: s6 - 32 = s6 & 0x1F; s6 < 32: 0xE0 | s6
            const uint32_t ge = (s6 >> 5) & 0x01010101u;
            st32(dst + 192 + 4 * g, (s6 & 0x1F1F1F1Fu) | ((ge ^ 0x01010101u) * 0xE0u));
        }
    } else {
        *reinterpret_cast<uint16_t *>(dst + 208) = ld_u16(src + 108);
    }
}

// 18 threads per source superblock {scales[16], qs[64], fp16 d, fp16 dmin} (84 bytes) -> superblock sb of the row's scale part and superblock
// nsb + sb of its minimum part.  Threads 0..15 as above (weight 128 n + 32 j + l is (qs[32 n + l] >> 2 j) & 3); thread 16: the scales; 17: d, dmin.
__global__ __launch_bounds__(256) void k_convert_q2_K_q6_K2(const uint8_t *__restrict__ W, int64_t w_row_bytes, int M, int nsb, uint8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * nsb * 18) return;
    const int pc = (int)(t % 18); const int64_t u = t / 18;
    const int row = (int)(u / nsb), sb = (int)(u % nsb);
    const uint8_t *src = W + (int64_t)row * w_row_bytes + (int64_t)sb * 84;
    uint8_t *da = out + ((int64_t)row * 2 * nsb + sb) * 210, *db = da + (int64_t)nsb * 210;
    auto st32 = [](uint8_t *p, uint32_t v) { *reinterpret_cast<uint16_t *>(p) = (uint16_t)v; *reinterpret_cast<uint16_t *>(p + 2) = (uint16_t)(v >> 16); };
    if (pc < 16) {
        const int n = pc >> 3, l = 4 * (pc & 7);
        const uint32_t q = ld_u32_a2(src + 16 + 32 * n + l);
        // q6 = q2 + 32 = 0x20 | q2: low nibble q2, bits 4-5 = 2.   q6 = 33 = 0x21: low nibble 1, bits 4-5 = 2
        st32(da + 64 * n + l, (q & 0x03030303u) | (((q >> 4) & 0x03030303u) << 4));
        st32(da + 64 * n + 32 + l, ((q >> 2) & 0x03030303u) | (((q >> 6) & 0x03030303u) << 4));
        st32(da + 128 + 32 * n + l, 0xAAAAAAAAu);
        st32(db + 64 * n + l, 0x11111111u); st32(db + 64 * n + 32 + l, 0x11111111u); st32(db + 128 + 32 * n + l, 0xAAAAAAAAu);
    } else if (pc == 16) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t s = ld_u32_a2(src + 4 * g), m = (s >> 4) & 0x0F0F0F0Fu;
            st32(da + 192 + 4 * g, s & 0x0F0F0F0Fu);
            // byte-wise -m for m in 0..15: 0 -> 0, else 0xF0 | (16 - m)
            const uint32_t tt = 0x10101010u - m, nz = ((m + 0x0F0F0F0Fu) >> 4) & 0x01010101u;
            st32(db + 192 + 4 * g, (tt & 0x0F0F0F0Fu) | (nz * 0xF0u));
        }
    } else {
        *reinterpret_cast<uint16_t *>(da + 208) = ld_u16(src + 80);
        *reinterpret_cast<uint16_t *>(db + 208) = ld_u16(src + 82);
    }
}

// IQ4_XS -> [h part | l part] as Q6_K: codebook value kv = 4 h + l with h = kv >> 2 in [-32, 28] and l = kv & 3, so
// w = d (ls - 32) kv = d (4 (ls - 32)) h + d (ls - 32) l: two Q6_K weights with int8 scales 4 (ls - 32) in [-128, 124] and ls - 32, q6 = h + 32 and
// l + 32, the same fp16 d.  d (ls - 32) has at most 17 significant bits, so both products and their sum are exact in fp32: the sum of the two
// parts' dequantize_row_q6_K equals dequantize_row_iq4_xs (src/ggml-quants.c:2454-2475) value for value.  18 threads per superblock as above:
// threads 0..15 (128-half n = pc >> 3, l = 4 (pc & 7) .. + 3): weight 128 n + 32 j + l is code (l & 15) (low codes for l < 16, high above) of
// sub-block 4 n + j; thread 16: the sixteen scales of each part (sub-block ib = 16-weight groups 2 ib, 2 ib + 1); thread 17: d.
// VAR (diagnosis of the round-3 instability, scripts/gpu_diag_iq4xs3.py): 0 = as shipped in round 3; 1 = 16-bit stores only (volatile: never merged into the
// 2-byte-aligned dword / dwordx4 stores hipcc makes of st32); 2 = codebook from a __constant__ table instead of the 64-bit shift LUT; 4 = every load waited
// for before the first use; combinations by OR
__device__ const int8_t k_iq4nl_values[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
template <int VAR>
__global__ __launch_bounds__(256) void k_convert_iq4_xs_q6_K2(const uint8_t *__restrict__ W, int64_t w_row_bytes, int M, int nsb, uint8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * nsb * 18) return;
    const int pc = (int)(t % 18); const int64_t u = t / 18;
    const int row = (int)(u / nsb), sb = (int)(u % nsb);
    const uint8_t *src = W + (int64_t)row * w_row_bytes + (int64_t)sb * 136;
    uint8_t *da = out + ((int64_t)row * 2 * nsb + sb) * 210, *db = da + (int64_t)nsb * 210;
    auto st32 = [](uint8_t *p, uint32_t v) {
        if constexpr ((VAR & 1) != 0) { *reinterpret_cast<volatile uint16_t *>(p) = (uint16_t)v; *reinterpret_cast<volatile uint16_t *>(p + 2) = (uint16_t)(v >> 16); }
        else { *reinterpret_cast<uint16_t *>(p) = (uint16_t)v; *reinterpret_cast<uint16_t *>(p + 2) = (uint16_t)(v >> 16); }
    };
    if (pc < 16) {
        const int n = pc >> 3, l = 4 * (pc & 7);
        uint32_t a[4], b[4];                                             // q6 of the h part / the l part for j = 0..3, one weight per byte
        uint32_t qq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) qq[j] = ld_u32_a2(src + 8 + 16 * (4 * n + j) + (l & 15));
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((VAR & 4) != 0) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3])::"memory"); }
#endif
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t q = qq[j];
            const uint32_t codes = l < 16 ? (q & 0x0F0F0F0Fu) : ((q >> 4) & 0x0F0F0F0Fu);
            uint32_t kv;
            if constexpr ((VAR & 2) != 0) {
                kv = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) kv |= (uint32_t)(uint8_t)k_iq4nl_values[(codes >> (8 * e)) & 0xFu] << (8 * e);
            } else kv = iq4nl_lut4(codes);
            a[j] = ((kv ^ 0x80808080u) >> 2) & 0x3F3F3F3Fu;              // (kv + 128) / 4 = (kv >> 2) + 32
            b[j] = (kv & 0x03030303u) | 0x20202020u;                     // (kv & 3) + 32
        }
        st32(da + 64 * n + l, (a[0] & 0x0F0F0F0Fu) | ((a[2] & 0x0F0F0F0Fu) << 4));
        st32(da + 64 * n + 32 + l, (a[1] & 0x0F0F0F0Fu) | ((a[3] & 0x0F0F0F0Fu) << 4));
        st32(da + 128 + 32 * n + l, ((a[0] >> 4) & 0x03030303u) | (((a[1] >> 4) & 0x03030303u) << 2) | (((a[2] >> 4) & 0x03030303u) << 4) | (((a[3] >> 4) & 0x03030303u) << 6));
        st32(db + 64 * n + l, (b[0] & 0x0F0F0F0Fu) | ((b[2] & 0x0F0F0F0Fu) << 4));
        st32(db + 64 * n + 32 + l, (b[1] & 0x0F0F0F0Fu) | ((b[3] & 0x0F0F0F0Fu) << 4));
        st32(db + 128 + 32 * n + l, 0xAAAAAAAAu);                        // bits 4-5 of every l-part q6 are 2
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((VAR & 8) != 0) { const uint8_t *keep = src + 8 + 64 * n + (l & 15); asm volatile("" ::"v"(keep)); }      // the loads' address registers stay untouched to the end
#endif
    } else if (pc == 16) {
        const uint32_t sh = ld_u16(src + 2);
#pragma unroll
        for (int ib = 0; ib < 8; ib++) {
            const int ls = (int)(((src[4 + (ib >> 1)] >> (4 * (ib & 1))) & 0xFu) | (((sh >> (2 * ib)) & 3u) << 4)) - 32;
            const uint32_t sa = (uint32_t)(4 * ls) & 0xFFu, sbv = (uint32_t)ls & 0xFFu;
            *reinterpret_cast<uint16_t *>(da + 192 + 2 * ib) = (uint16_t)(sa | (sa << 8));
            *reinterpret_cast<uint16_t *>(db + 192 + 2 * ib) = (uint16_t)(sbv | (sbv << 8));
        }
    } else {
        *reinterpret_cast<uint16_t *>(da + 208) = ld_u16(src);
        *reinterpret_cast<uint16_t *>(db + 208) = ld_u16(src);
    }
}

size_t cdna4_convert_weights_bytes(int type, int64_t M, int64_t K) {
    if (type == CDNA4_Q2_K || type == CDNA4_IQ4_XS) return (size_t)M * 2 * (K / 256) * 210;
    if (type == CDNA4_Q5_0 || type == CDNA4_IQ4_NL) return (size_t)M * (K / 32) * 34;
    if (type == CDNA4_Q4_1 || type == CDNA4_Q5_1) return (size_t)M * 2 * (K / 32) * 34;
    if (type == CDNA4_Q3_K) return (size_t)M * (K / 256) * 210;
    return 0;
}
int cdna4_convert_weights_target(int type) {
    if (type == CDNA4_Q5_0 || type == CDNA4_IQ4_NL || type == CDNA4_Q4_1 || type == CDNA4_Q5_1) return CDNA4_Q8_0;
    return (type == CDNA4_Q3_K || type == CDNA4_Q2_K || type == CDNA4_IQ4_XS) ? CDNA4_Q6_K : -1;
}
int cdna4_convert_weights_kmul(int type) { return (type == CDNA4_Q2_K || type == CDNA4_Q4_1 || type == CDNA4_Q5_1 || type == CDNA4_IQ4_XS) ? 2 : 1; }       // columns of the result per column of the source

// W [M rows, w_row_bytes apart] of `type` -> `out` (contiguous rows of the target format); 2-byte aligned source rows
int cdna4_launch_convert_weights(int type, const uint8_t *W, int64_t w_row_bytes, int64_t M, int64_t K, uint8_t *out, hipStream_t st) {
    if (M <= 0 || K <= 0) return 0;
    if (((uintptr_t)W | (uintptr_t)w_row_bytes | (uintptr_t)out) & 1) return cdna4_set_error_msg("convert_weights: rows must be 2-byte aligned");
    if (type == CDNA4_Q5_0) {
        if (K % 32) return cdna4_set_error_msg("convert_weights: K must be a multiple of 32");
        const int64_t n = M * (K / 32);
        hipLaunchKernelGGL(k_convert_q5_0_q8_0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, w_row_bytes, (int)M, (int)(K / 32), out);
    } else if (type == CDNA4_Q4_1 || type == CDNA4_Q5_1 || type == CDNA4_IQ4_NL) {
        if (K % 32) return cdna4_set_error_msg("convert_weights: K must be a multiple of 32");
        const int64_t n = M * (K / 32);
        const dim3 grid((unsigned)((n + 255) / 256));
        if (type == CDNA4_Q4_1) hipLaunchKernelGGL(k_convert_q41_q8_0x2<false>, grid, dim3(256), 0, st, W, w_row_bytes, (int)M, (int)(K / 32), out);
        else if (type == CDNA4_Q5_1) hipLaunchKernelGGL(k_convert_q41_q8_0x2<true>, grid, dim3(256), 0, st, W, w_row_bytes, (int)M, (int)(K / 32), out);
        else hipLaunchKernelGGL(k_convert_iq4_nl_q8_0, grid, dim3(256), 0, st, W, w_row_bytes, (int)M, (int)(K / 32), out);
    } else if (type == CDNA4_Q3_K) {
        if (K % 256) return cdna4_set_error_msg("convert_weights: K must be a multiple of 256");
        const int64_t n = M * (K / 256) * 18;
        hipLaunchKernelGGL(k_convert_q3_K_q6_K, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, w_row_bytes, (int)M, (int)(K / 256), out);
    } else if (type == CDNA4_IQ4_XS) {
        if (K % 256) return cdna4_set_error_msg("convert_weights: K must be a multiple of 256");
        const int64_t n = M * (K / 256) * 18;
        // (the diagnosis variants of round 3's instability — VAR 0 / 1 / 2 / 3 / 5 / 7 / 8, CDNA4_DIAG_CONV — are no longer instantiated: VAR 4, every load waited for before its first
        //  use, is what ships; profiles/r04/iq4xs_rootcause.txt has the sweep.  Since round 5 a resident image is also built twice and compared at load: gemm_q_mfma.hip.)
        hipLaunchKernelGGL(k_convert_iq4_xs_q6_K2<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, w_row_bytes, (int)M, (int)(K / 256), out);
    } else if (type == CDNA4_Q2_K) {
        if (K % 256) return cdna4_set_error_msg("convert_weights: K must be a multiple of 256");
        const int64_t n = M * (K / 256) * 18;
        hipLaunchKernelGGL(k_convert_q2_K_q6_K2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, w_row_bytes, (int)M, (int)(K / 256), out);
    } else return cdna4_set_error_msg("convert_weights: no exact target format for this type");
    CDNA4_CHECK_LAUNCH();
    return 0;
}
