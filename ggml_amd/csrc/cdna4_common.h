// cdna4_common.h — shared device helpers for the MI355X (gfx950, wave64) ggml kernels.
// Block formats follow the reference's normative layout (src/ggml-common.h:161-328); nothing here is
// derived from src/ggml-cuda.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CDNA4_WAVE 64
#define QK_K 256

// ggml type ids (include/ggml.h:351-390)
enum cdna4_type : int {
    CDNA4_F32 = 0, CDNA4_F16 = 1, CDNA4_Q4_0 = 2, CDNA4_Q8_0 = 8,
    CDNA4_Q4_K = 12, CDNA4_Q5_K = 13, CDNA4_Q6_K = 14, CDNA4_Q8_K = 15, CDNA4_I32 = 26,
    // MUL_MAT / MUL_MAT_ID through the GEMV units of gemv_q.hip; prefill GEMM of Q5_0 / Q3_K through convert_w.hip; to_float in ops.hip
    CDNA4_Q5_0 = 6, CDNA4_Q2_K = 10, CDNA4_Q3_K = 11,
    // the same (GEMV units on Q8_1 activations; prefill GEMM as [d q | m 1] in Q8_0 against a doubled activation image)
    CDNA4_Q4_1 = 3, CDNA4_Q5_1 = 7,
    // the same: non-linear 4-bit codebook (kvalues_iq4nl, src/ggml-quants.c:2434), Q8_0 activations; prefill GEMM as Q8_0 (q8 = codebook value)
    CDNA4_IQ4_NL = 20,
    // the same with 256-weight superblocks and 6-bit sub-block scales (Q8_K activations); prefill GEMM as Q6_K with 2 K columns (codebook value =
    // 4 h + l: the h part with int8 scale 4 (ls - 32), the l part with ls - 32)
    CDNA4_IQ4_XS = 23,
    // K / V of FLASH_ATTN_EXT only (a bf16 KV cache; written out as fp16 like the quantized ones, ops.hip: k_q_to_f16_dense)
    CDNA4_BF16 = 30,
};
// weight types whose CPU vec_dot runs on Q8_1 activations (type_traits_cpu[].vec_dot_type, src/ggml-cpu/ggml-cpu.c:271-296): the activation
// workspace then carries, in the place of the Q8_K bsums, one fp32 per 32-block holding s = fp16(d * sum of the quants) (block_q8_1.s)
constexpr bool cdna4_is_q81(int t) { return t == CDNA4_Q4_1 || t == CDNA4_Q5_1; }

// bytes per block / weights per block
template <int T> struct QT;
template <> struct QT<CDNA4_Q4_0> { static constexpr int BYTES = 18,  QK = 32;  static constexpr bool KQ = false; };
template <> struct QT<CDNA4_Q8_0> { static constexpr int BYTES = 34,  QK = 32;  static constexpr bool KQ = false; };
template <> struct QT<CDNA4_Q4_K> { static constexpr int BYTES = 144, QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q5_K> { static constexpr int BYTES = 176, QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q6_K> { static constexpr int BYTES = 210, QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q5_0> { static constexpr int BYTES = 22,  QK = 32;  static constexpr bool KQ = false; };
template <> struct QT<CDNA4_Q2_K> { static constexpr int BYTES = 84,  QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q3_K> { static constexpr int BYTES = 110, QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q4_1> { static constexpr int BYTES = 20,  QK = 32;  static constexpr bool KQ = false; };
template <> struct QT<CDNA4_Q5_1> { static constexpr int BYTES = 24,  QK = 32;  static constexpr bool KQ = false; };
template <> struct QT<CDNA4_IQ4_NL> { static constexpr int BYTES = 18, QK = 32;  static constexpr bool KQ = false; };
template <> struct QT<CDNA4_IQ4_XS> { static constexpr int BYTES = 136, QK = 256; static constexpr bool KQ = true; };
// library-private re-layouts of the 2-byte-aligned formats into 16-byte-aligned 256-weight superblocks, produced per call
// into scratch by gemm_q_mfma.hip's repack kernels so that the LDS-DMA pipeline (16-byte pieces) can stage them:
//   Q4_0R 144 B: fp16 d[8] | 4 x 32 B nibbles in Q4_K order (byte l of group g: low = k 64g+l, high = k 64g+32+l)
//   Q8_0R 272 B: fp16 d[8] | 256 int8 in k order
//   Q6_KR 224 B: fp16 d, 14 B pad | int8 scales[16] | ql[128] | qh[64]
enum : int { CDNA4_Q4_0R = 102, CDNA4_Q8_0R = 108, CDNA4_Q6_KR = 114 };
// RESIDENT-only re-layout of Q6_K for k_gemm_r8 (round 5; cdna4_resident_*): the sixteen sub-block scales already multiplied by d, the 6-bit quants widened to int8 —
//   Q6_K8 288 B: fp16 s[16] (s_i = fp16(d * scales[i])) | 256 int8 (q - 32) in k order          (the product s_i * (q - 32) is the one every Q6_K GEMM kernel here takes)
enum : int { CDNA4_Q6_K8 = 115 };
// "staged" forms: never in HBM — the loader waves of k_gemm_kq_w12 read the ORIGINAL 2-byte-aligned blocks and write these
// 128-k stage rows straight into the LDS ring (re-layout while staging; no repack kernel, no scratch copy):
//   Q4_0S 80 B: fp16 d[4], 8 B pad | 2 x 32 B nibbles in Q4_K order       Q8_0S 144 B: fp16 d[4], pad | 2 x 64 int8
//   Q6_KS 144 B: fp16 d, pad | int8 scales[8], pad | ql[64] | qh[32] | pad   (one 128-weight half of a superblock)
enum : int { CDNA4_Q4_0S = 202, CDNA4_Q8_0S = 208, CDNA4_Q6_KS = 214 };
template <> struct QT<CDNA4_Q4_0S> { static constexpr int BYTES = 8 * 18, QK = 256; static constexpr bool KQ = true; };    // BYTES = source bytes per 256 weights
template <> struct QT<CDNA4_Q8_0S> { static constexpr int BYTES = 8 * 34, QK = 256; static constexpr bool KQ = false; };
template <> struct QT<CDNA4_Q6_KS> { static constexpr int BYTES = 210, QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q4_0R> { static constexpr int BYTES = 144, QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q8_0R> { static constexpr int BYTES = 272, QK = 256; static constexpr bool KQ = false; };
template <> struct QT<CDNA4_Q6_KR> { static constexpr int BYTES = 224, QK = 256; static constexpr bool KQ = true; };
template <> struct QT<CDNA4_Q6_K8> { static constexpr int BYTES = 288, QK = 256; static constexpr bool KQ = true; };

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// loads with a stated (possibly small) alignment: the compiler picks the widest legal instruction
struct __attribute__((packed, aligned(2))) u32_a2 { uint32_t v; };
struct __attribute__((packed, aligned(2))) u16_a2 { uint16_t v; };
__device__ __forceinline__ uint32_t ld_u32_a2(const uint8_t *p) { return reinterpret_cast<const u32_a2 *>(p)->v; }
__device__ __forceinline__ uint16_t ld_u16(const uint8_t *p) { return *reinterpret_cast<const uint16_t *>(p); }
__device__ __forceinline__ u32x4 ld_u32x4(const void *p) { return *reinterpret_cast<const u32x4 *>(p); }

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(half_t, h); }
// fp32 -> fp16 bits, RNE, of the fp32 VALUE f.  The empty asm keeps f opaque: hipcc otherwise folds a producing multiply into the conversion
// (v_fma_mixlo_f16 a, b, 0 — ONE rounding of the exact product) where the CPU rounds the product to fp32 first and that to fp16; the two
// differ by one fp16 ulp when the fp32 rounding lands on an fp16 tie.  Seen on MI355X as block_q8_1.s = fp16(d * sum) one ulp off the
// reference in one block of 256 (round 2's only hardware parity failure); the CPU emulator, which converts in two steps, could not see it.
__device__ __forceinline__ float opaque_f32(float f) { asm("" : "+v"(f)); return f; }
__device__ __forceinline__ uint16_t f2h_bits(float f) { return __builtin_bit_cast(uint16_t, (half_t)opaque_f32(f)); }
// four 4-bit codes (one per byte, 0..15) -> their four int8 codebook values kvalues_iq4nl[] = {-127, -104, -83, -65, -49, -35, -22, -10,
// 1, 13, 25, 38, 53, 69, 89, 113} (src/ggml-quants.c:2434), packed the same way.  Plain shifts of the table held in two 64-bit constants.
__device__ __forceinline__ uint32_t iq4nl_lut4(uint32_t codes) {
    constexpr uint64_t T_LO = 0xF6EADDCFBFAD9881ull, T_HI = 0x7159453526190D01ull;         // entries 0..7 / 8..15, entry i in byte i & 7
    uint32_t r = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t c = (codes >> (8 * e)) & 0xFu;
        const uint64_t t = (c & 8u) ? T_HI : T_LO;
        r |= ((uint32_t)(t >> (8 * (c & 7u))) & 0xFFu) << (8 * e);
    }
    return r;
}

// ---- cross-lane moves on the VALU (DPP) instead of the LDS crossbar: __shfl_xor compiles to ds_bpermute_b32 (~100 cycles issue -> use; a
// 16-lane butterfly of three values is a dependent chain of 12 of them, ~0.5 us in the decode kernel's quantizer).  gfx950's DPP has no xor-4 /
// xor-8 pattern, but for ALL-REDUCES (every lane wants the group's result) quad_perm + row_half_mirror + row_mirror do: after the two
// quad steps the lanes of a quad agree, and the mirrors pair every quad with the other one(s) of its half / row.
//   ctrl: 0xB1 quad_perm [1,0,3,2]   0x4E quad_perm [2,3,0,1]   0x141 row_half_mirror   0x140 row_mirror   0x142 row_bcast:15   0x143 row_bcast:31
#if defined(__HIPCC__)
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ int dpp_i32(int v, int old = 0) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF); }
#else   // host builds of the kernel sources (tools/emul): the same lane mapping through the emulated wave shuffle
template <int CTRL, int ROW_MASK = 0xF> static inline int dpp_i32(int v, int old = 0) {
    const int l = (int)(threadIdx.x & 63), r = l >> 4, i = l & 15;
    int src = l; bool writes = (ROW_MASK >> r) & 1;
    if (CTRL < 0x100) src = (l & ~3) | ((CTRL >> (2 * (l & 3))) & 3);
    else if (CTRL == 0x141) src = (l & ~7) | (7 - (l & 7));
    else if (CTRL == 0x140) src = (l & ~15) | (15 - i);
    else if (CTRL == 0x142) { src = ((r - 1) << 4) | 15; writes = writes && r > 0; }
    else if (CTRL == 0x143) { src = 31; writes = writes && r >= 2; }
    const int got = emu_shfl_idx(v, src < 0 ? 0 : src);
    return writes ? got : old;
}
#endif
// LDS traffic between the lanes of ONE wave needs no barrier on the GPU (a wave's LDS operations execute in order) — only the compiler must
// not move the reads over the writes; on the host build every lane is an OS thread and has to wait for its wave
#if defined(__HIPCC__)
#define CDNA4_WAVE_LDS_SYNC() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#else
#define CDNA4_WAVE_LDS_SYNC() emu::wave_sync()
#endif
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_f32(float v, float old = 0.f) {
    return __builtin_bit_cast(float, dpp_i32<CTRL, ROW_MASK>(__builtin_bit_cast(int, v), __builtin_bit_cast(int, old)));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum of v over the 64 lanes, returned in every lane: rows of 16 by DPP all-reduce, then row_bcast:15 / :31 carry the row totals into lane 63.
// ALL 64 lanes must be active (v_readlane of an inactive lane reads a stale register).  Used by the decode kernels (gemv_q.hip).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);        // every lane: its row's sum
    v += dpp_f32<0x142, 0xA>(v);                                                                         // rows 1, 3 += rows 0, 2
    v += dpp_f32<0x143, 0xC>(v);                                                                         // rows 2, 3 += row 1 (= rows 0 + 1)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// true in every lane iff the predicate holds in some ACTIVE lane of the wave
__device__ __forceinline__ bool wave_any(bool pred) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(pred) != 0;
#else
    int v = pred ? 1 : 0;                                  // (tools/emul: all 64 lanes of the wave execute this)
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v != 0;
#endif
}
// bit l set iff the predicate holds in lane l (all 64 lanes of the wave execute this)
__device__ __forceinline__ uint64_t wave_ballot(bool pred) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(pred);
#else
    const int l = (int)(threadIdx.x & 63);                 // (tools/emul: OR-reduction over the emulated shuffle)
    int lo = (pred && l < 32) ? (int)(1u << l) : 0, hi = (pred && l >= 32) ? (int)(1u << (l - 32)) : 0;
    for (int o = 32; o > 0; o >>= 1) { lo |= __shfl_xor(lo, o, 64); hi |= __shfl_xor(hi, o, 64); }
    return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 6-bit scale/min pairs of Q4_K / Q5_K: get_scale_min_k4, src/ggml-quants.c:631-638.
// sc[0..2] are the 12 scale bytes as three little-endian dwords.
__device__ __forceinline__ uint32_t k4_byte(const uint32_t sc[3], int j) { return (sc[j >> 2] >> (8 * (j & 3))) & 0xFF; }
__device__ __forceinline__ void k4_scale_min(const uint32_t sc[3], int j, uint32_t &d, uint32_t &m) {
    if (j < 4) { d = k4_byte(sc, j) & 63; m = k4_byte(sc, j + 4) & 63; }
    else { d = (k4_byte(sc, j + 4) & 0xF) | ((k4_byte(sc, j - 4) >> 6) << 4); m = (k4_byte(sc, j + 4) >> 4) | ((k4_byte(sc, j) >> 6) << 4); }
}
// same with a run-time j on registers (select chains, no scratch)
__device__ __forceinline__ uint32_t k4_byte_rt(uint32_t a, uint32_t b, uint32_t c, int j) {
    const uint32_t w = j < 4 ? a : (j < 8 ? b : c);
    return (w >> (8 * (j & 3))) & 0xFF;
}
__device__ __forceinline__ void k4_scale_min_rt(uint32_t a, uint32_t b, uint32_t c, int j, int &d, int &m) {
    if (j < 4) { d = k4_byte_rt(a, b, c, j) & 63; m = k4_byte_rt(a, b, c, j + 4) & 63; }
    else { d = (k4_byte_rt(a, b, c, j + 4) & 0xF) | ((k4_byte_rt(a, b, c, j - 4) >> 6) << 4);
           m = (k4_byte_rt(a, b, c, j + 4) >> 4) | ((k4_byte_rt(a, b, c, j) >> 6) << 4); }
}

// the (scale, min) pairs of the TWO sub-blocks 2 g and 2 g + 1 of a Q4_K / Q5_K superblock — what a decode unit (one 64-weight group g) needs — from the 12 scale bytes
// a, b, c: bit fields at ONE run-time shift (16 (g & 1)) instead of the byte-select chains of k4_scale_min_rt (about 20 VALU for the four values instead of about 50; the decode
// kernels are instruction-bound, profiles/r04/pmc_decode_counters.txt).  Same values as get_scale_min_k4(2 g | 2 g + 1), src/ggml-quants.c:631-638: for j < 4 the low six
// bits of bytes j / j + 4; for j >= 4 the nibbles of byte j + 4 topped by the two high bits of bytes j - 4 / j.
__device__ __forceinline__ void k4_scale_min_pair(uint32_t a, uint32_t b, uint32_t c, int g, int &sc0, int &m0, int &sc1, int &m1) {
    const uint32_t sh = 16u * (uint32_t)(g & 1);
    const uint32_t ya = a >> sh, yb = b >> sh, yc = c >> sh;             // bytes (2 g & 3) and (2 g & 3) + 1 in the low half
    if (g < 2) {
        sc0 = (int)(ya & 63u); m0 = (int)(yb & 63u); sc1 = (int)((ya >> 8) & 63u); m1 = (int)((yb >> 8) & 63u);
    } else {
        sc0 = (int)((yc & 0xFu) | (((ya >> 6) & 3u) << 4)); m0 = (int)(((yc >> 4) & 0xFu) | (((yb >> 6) & 3u) << 4));
        sc1 = (int)(((yc >> 8) & 0xFu) | (((ya >> 14) & 3u) << 4)); m1 = (int)(((yc >> 12) & 0xFu) | (((yb >> 14) & 3u) << 4));
    }
}

#define CDNA4_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return cdna4_set_error(e_, __FILE__, __LINE__); } while (0)
int cdna4_set_error(hipError_t e, const char *file, int line);
int cdna4_set_error_msg(const char *msg);
