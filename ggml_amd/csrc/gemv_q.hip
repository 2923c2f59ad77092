// gemv_q.hip — single-token decode path: Y[c][m] = sum_k W[m][k] * X[c][k] for 1..8 activation columns,
// computed the way the reference CPU backend computes it (integer block dot products on Q8_K / Q8_0
// quantized activations, fp32 scales applied per block) so results agree with ggml-cpu to fp32
// summation order.  Follows ggml_vec_dot_q4_K_q8_K / _q5_K_ / _q6_K_ / _q4_0_q8_0 / _q8_0_q8_0
// (src/ggml-cpu/ggml-cpu-quants.c:6137-6193, 6769-6830, 7425-7467, 2293-2310, 3335-...).
//
// MI355X mapping: HBM-bound.  One wave per weight row, lane = one 64-weight (K-quants) or 32-weight
// (Q4_0/Q8_0) unit, so a wave issues 16-byte loads over whole contiguous superblocks; all of a row's
// bytes are in flight before the first v_dot4_i32_i8; 64-lane butterfly reduction at the end.
// Also serves MUL_MAT_ID (per-column expert base, ids read on the device — no host sync).
#include "cdna4_common.h"
#include "cdna4_kernels.h"
#include "quantize_dev.h"
#include "epilogue.h"
#include "gemm_q_hw.h"
#include <stdlib.h>
int cdna4_gemm_cu_count();                                          // gemm_q_mfma.hip

__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

template <int TYPE, int NB> struct Unit;

// The quantized activations a unit meets: in global memory (cdna4_gemv_args: the workspace ggml_cdna4_prepare_act filled) or in LDS (lds_act:
// the one-launch and staged forms).  In LDS a lane reads the 16-byte chunks of ITS unit, i.e. lanes stride 64 (K-quants) or 32 bytes: 4-way /
// 2-way bank conflicts on every ds_read_b128 (the 16 lanes of a read group would share a quarter / half of the 64 banks).  So the int8 row
// is stored with its chunk index XOR-swizzled, c -> c ^ ((c >> 4) & 3): conflict-free for both unit sizes under the read groups of
// MI355X_MICROARCH.md (checked by brute force); writers (quantizer, staging copy) and readers go through lds_swz().
struct lds_act { const int8_t *qs; const float *d; const int16_t *bsums; int K; };
__device__ __forceinline__ int lds_swz(int byte_off) { const int c = byte_off >> 4; return ((c ^ ((c >> 4) & 3)) << 4) | (byte_off & 15); }
template <typename ACT> __device__ __forceinline__ u32x4 act_ld16(const ACT &, const int8_t *row, int off) { return ld_u32x4(row + off); }
__device__ __forceinline__ u32x4 act_ld16(const lds_act &, const int8_t *row, int off) { return ld_u32x4(row + lds_swz(off)); }

// Every unit is split into load() — all of the unit's weight bytes into registers, no activation dependence — and
// mac() — the integer dots against the quantized activations.  dot() = mac(load()).  The fused decode kernel calls
// load() BEFORE it quantizes the activation row, so the weight stream's HBM latency overlaps the quantizer.

// ---- Q4_K: unit = (superblock, 64-group g): 32 bytes of nibbles, low -> k 64g+l, high -> 64g+32+l ---------
template <int NB> struct Unit<CDNA4_Q4_K, NB> {
    static constexpr int UK = 64;
    struct W { u32x4 hdr, q0, q1; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const int sb = u >> 2, g = u & 3;
        const uint8_t *blk = wrow + (int64_t)sb * 144;
        W w; w.hdr = ld_u32x4(blk); w.q0 = ld_u32x4(blk + 16 + 32 * g); w.q1 = ld_u32x4(blk + 32 + 32 * g);
        return w;
    }
    // (non-temporal weight loads were measured slower here: 6.06 vs 5.58 us cold at 4096x4096)
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, g = u & 3;
        const u32x4 hdr = wr.hdr;
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        int sc_lo, m_lo, sc_hi, m_hi;
        k4_scale_min_pair(hdr.y, hdr.z, hdr.w, g, sc_lo, m_lo, sc_hi, m_hi);
        const uint32_t w[8] = {wr.q0.x, wr.q0.y, wr.q0.z, wr.q0.w, wr.q1.x, wr.q1.y, wr.q1.z, wr.q1.w};
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = sb * 256 + 64 * g;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16), y2 = act_ld16(a, yr, yo + 32), y3 = act_ld16(a, yr, yo + 48);
            const uint32_t yl[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const uint32_t yh[8] = {y2.x, y2.y, y2.z, y2.w, y3.x, y3.y, y3.z, y3.w};
            int sl = 0, sh = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { sl = dot4(w[i] & 0x0F0F0F0Fu, yl[i], sl); sh = dot4((w[i] >> 4) & 0x0F0F0F0Fu, yh[i], sh); }
            const u32x2 bs = *reinterpret_cast<const u32x2 *>(a.bsums + (int64_t)col[c] * (a.K / 16) + sb * 16 + 4 * g);
            const int blo = (int)(int16_t)(bs.x & 0xFFFF) + (int)(int16_t)(bs.x >> 16);
            const int bhi = (int)(int16_t)(bs.y & 0xFFFF) + (int)(int16_t)(bs.y >> 16);
            const float yd = a.d[(int64_t)col[c] * (a.K / 256) + sb];
            acc[c] += (d * yd) * (float)(sc_lo * sl + sc_hi * sh) - (dmin * yd) * (float)(m_lo * blo + m_hi * bhi);
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- Q5_K: Q4_K plus a fifth bit per weight from qh[l] bit 2g (low) / 2g+1 (high) ---------------------------
template <int NB> struct Unit<CDNA4_Q5_K, NB> {
    static constexpr int UK = 64;
    struct W { u32x4 hdr, h0, h1, q0, q1; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const int sb = u >> 2, g = u & 3;
        const uint8_t *blk = wrow + (int64_t)sb * 176;
        W w; w.hdr = ld_u32x4(blk); w.h0 = ld_u32x4(blk + 16); w.h1 = ld_u32x4(blk + 32);
        w.q0 = ld_u32x4(blk + 48 + 32 * g); w.q1 = ld_u32x4(blk + 64 + 32 * g);
        return w;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, g = u & 3;
        const u32x4 hdr = wr.hdr;
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        int sc_lo, m_lo, sc_hi, m_hi;
        k4_scale_min_pair(hdr.y, hdr.z, hdr.w, g, sc_lo, m_lo, sc_hi, m_hi);
        const uint32_t qh[8] = {wr.h0.x, wr.h0.y, wr.h0.z, wr.h0.w, wr.h1.x, wr.h1.y, wr.h1.z, wr.h1.w};
        const uint32_t w[8] = {wr.q0.x, wr.q0.y, wr.q0.z, wr.q0.w, wr.q1.x, wr.q1.y, wr.q1.z, wr.q1.w};
        uint32_t wl[8], wh[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            wl[i] = (w[i] & 0x0F0F0F0Fu) | (((qh[i] >> (2 * g)) & 0x01010101u) << 4);
            wh[i] = ((w[i] >> 4) & 0x0F0F0F0Fu) | (((qh[i] >> (2 * g + 1)) & 0x01010101u) << 4);
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = sb * 256 + 64 * g;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16), y2 = act_ld16(a, yr, yo + 32), y3 = act_ld16(a, yr, yo + 48);
            const uint32_t yl[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const uint32_t yh[8] = {y2.x, y2.y, y2.z, y2.w, y3.x, y3.y, y3.z, y3.w};
            int sl = 0, sh = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { sl = dot4(wl[i], yl[i], sl); sh = dot4(wh[i], yh[i], sh); }
            const u32x2 bs = *reinterpret_cast<const u32x2 *>(a.bsums + (int64_t)col[c] * (a.K / 16) + sb * 16 + 4 * g);
            const int blo = (int)(int16_t)(bs.x & 0xFFFF) + (int)(int16_t)(bs.x >> 16);
            const int bhi = (int)(int16_t)(bs.y & 0xFFFF) + (int)(int16_t)(bs.y >> 16);
            const float yd = a.d[(int64_t)col[c] * (a.K / 256) + sb];
            acc[c] += (d * yd) * (float)(sc_lo * sl + sc_hi * sh) - (dmin * yd) * (float)(m_lo * blo + m_hi * bhi);
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- Q6_K: unit = (superblock, half n, 16-lane slice lb): 16 values of l -> 4 x 16 weights at
//      k = 128n + 32*quad + 16lb + i, int8 scale per 16 (src/ggml-quants.c:1690-1719) -----------------------
template <int NB> struct Unit<CDNA4_Q6_K, NB> {
    static constexpr int UK = 64;
    struct W { uint32_t q[4][4]; int sc[4]; float d; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const int sb = u >> 2, n = (u >> 1) & 1, lb = u & 1;
        const uint8_t *blk = wrow + (int64_t)sb * 210;                       // 2-byte aligned only
        const uint8_t *ql_a = blk + 64 * n + 16 * lb, *ql_b = ql_a + 32, *qhp = blk + 128 + 32 * n + 16 * lb;
        const int8_t *scp = reinterpret_cast<const int8_t *>(blk + 192 + 8 * n + lb);
        W w; w.d = h2f(ld_u16(blk + 208));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t la = ld_u32_a2(ql_a + 4 * i), lv = ld_u32_a2(ql_b + 4 * i), hq = ld_u32_a2(qhp + 4 * i);
            w.q[0][i] = (la & 0x0F0F0F0Fu) | ((hq & 0x03030303u) << 4);
            w.q[1][i] = (lv & 0x0F0F0F0Fu) | (((hq >> 2) & 0x03030303u) << 4);
            w.q[2][i] = ((la >> 4) & 0x0F0F0F0Fu) | (((hq >> 4) & 0x03030303u) << 4);
            w.q[3][i] = ((lv >> 4) & 0x0F0F0F0Fu) | (((hq >> 6) & 0x03030303u) << 4);
        }
        w.sc[0] = scp[0]; w.sc[1] = scp[2]; w.sc[2] = scp[4]; w.sc[3] = scp[6];
        return w;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, n = (u >> 1) & 1, lb = u & 1;
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = sb * 256 + 128 * n + 16 * lb;
            const int16_t *bs = a.bsums + (int64_t)col[c] * (a.K / 16) + sb * 16 + 8 * n + lb;
            int isum = 0;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const u32x4 yv = act_ld16(a, yr, yo + 32 * qd);
                int s = dot4(wr.q[qd][0], yv.x, 0); s = dot4(wr.q[qd][1], yv.y, s); s = dot4(wr.q[qd][2], yv.z, s); s = dot4(wr.q[qd][3], yv.w, s);
                isum += wr.sc[qd] * (s - 32 * (int)bs[2 * qd]);             // sum (q-32)*y = sum q*y - 32*bsum
            }
            const float yd = a.d[(int64_t)col[c] * (a.K / 256) + sb];
            acc[c] += (wr.d * yd) * (float)isum;
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- formats beyond the five headline ones: Q5_0 / Q2_K / Q3_K (hardware-verified in round 2), Q4_1 / Q5_1 / IQ4_NL / IQ4_XS
// ---- (written against the CPU oracle, verified on the CPU emulator: tools/emul/gemv_emul) ----------------------------
// bits 0..3 of x -> bit 0 of bytes 0..3
__device__ __forceinline__ uint32_t spread4(uint32_t x) { return ((x & 0xFu) * 0x00204081u) & 0x01010101u; }

// ---- Q5_0: 22-byte block {fp16 d, qh[4], qs[16]}: Q4_0 plus a fifth bit (bit j of qh: weight j; bit j+16: weight j+16), value q-16
template <int NB> struct Unit<CDNA4_Q5_0, NB> {
    static constexpr int UK = 32;
    struct W { float d; uint32_t qh; uint32_t w[4]; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const uint8_t *blk = wrow + (int64_t)u * 22;
        W r; r.d = h2f(ld_u16(blk)); r.qh = ld_u32_a2(blk + 2);
#pragma unroll
        for (int i = 0; i < 4; i++) r.w[i] = ld_u32_a2(blk + 6 + 4 * i);
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        uint32_t wl[4], wh[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            wl[i] = (wr.w[i] & 0x0F0F0F0Fu) | (spread4(wr.qh >> (4 * i)) << 4);
            wh[i] = ((wr.w[i] >> 4) & 0x0F0F0F0Fu) | (spread4(wr.qh >> (16 + 4 * i)) << 4);
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = u * 32;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16);
            const uint32_t yl[4] = {y0.x, y0.y, y0.z, y0.w}, yh[4] = {y1.x, y1.y, y1.z, y1.w};
            int s = 0, ys = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s = dot4(wl[i], yl[i], s); s = dot4(wh[i], yh[i], s);
                ys = dot4(0x01010101u, yl[i], ys); ys = dot4(0x01010101u, yh[i], ys);
            }
            acc[c] += (float)(s - 16 * ys) * wr.d * a.d[(int64_t)col[c] * (a.K / 32) + u];
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- Q4_1 / Q5_1: Q8_1 activations (vec_dot_q4_1_q8_1 / q5_1_q8_1, src/ggml-cpu/ggml-cpu-quants.c:2585-2601, 3309-3331):
// ---- per 32-block (d_w * d_y) * sum(q_w q_y) + m_w * s_y, s_y = fp16(d * sum q_y) written by the activation quantizer where the
// ---- K-quants keep their bsums (one fp32 per block: act_s())
template <typename ACT> __device__ __forceinline__ float act_s(const ACT &a, int64_t col, int u) { return reinterpret_cast<const float *>(a.bsums)[col * (a.K / 32) + u]; }

// Q4_1: 20-byte block {fp16 d, fp16 m, qs[16]}: nibble j -> k j (low), j+16 (high), value q d + m
template <int NB> struct Unit<CDNA4_Q4_1, NB> {
    static constexpr int UK = 32;
    struct W { float d, m; uint32_t w[4]; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const uint8_t *blk = wrow + (int64_t)u * 20;
        W r; r.d = h2f(ld_u16(blk)); r.m = h2f(ld_u16(blk + 2));
#pragma unroll
        for (int i = 0; i < 4; i++) r.w[i] = ld_u32_a2(blk + 4 + 4 * i);
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = u * 32;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16);
            const uint32_t yl[4] = {y0.x, y0.y, y0.z, y0.w}, yh[4] = {y1.x, y1.y, y1.z, y1.w};
            int s = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { s = dot4(wr.w[i] & 0x0F0F0F0Fu, yl[i], s); s = dot4((wr.w[i] >> 4) & 0x0F0F0F0Fu, yh[i], s); }
            acc[c] += (wr.d * a.d[(int64_t)col[c] * (a.K / 32) + u]) * (float)s + wr.m * act_s(a, col[c], u);
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// Q5_1: 24-byte block {fp16 d, fp16 m, qh[4], qs[16]}: Q4_1 plus a fifth bit (bit j of qh: weight j; bit j+16: weight j+16)
template <int NB> struct Unit<CDNA4_Q5_1, NB> {
    static constexpr int UK = 32;
    struct W { float d, m; uint32_t qh; uint32_t w[4]; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const uint8_t *blk = wrow + (int64_t)u * 24;
        W r; r.d = h2f(ld_u16(blk)); r.m = h2f(ld_u16(blk + 2)); r.qh = ld_u32_a2(blk + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) r.w[i] = ld_u32_a2(blk + 8 + 4 * i);
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        uint32_t wl[4], wh[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            wl[i] = (wr.w[i] & 0x0F0F0F0Fu) | (spread4(wr.qh >> (4 * i)) << 4);
            wh[i] = ((wr.w[i] >> 4) & 0x0F0F0F0Fu) | (spread4(wr.qh >> (16 + 4 * i)) << 4);
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = u * 32;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16);
            const uint32_t yl[4] = {y0.x, y0.y, y0.z, y0.w}, yh[4] = {y1.x, y1.y, y1.z, y1.w};
            int s = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { s = dot4(wl[i], yl[i], s); s = dot4(wh[i], yh[i], s); }
            acc[c] += (wr.d * a.d[(int64_t)col[c] * (a.K / 32) + u]) * (float)s + wr.m * act_s(a, col[c], u);
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// IQ4_NL: 18-byte block {fp16 d, qs[16]} — Q4_0's layout with the 4-bit codes looked up in the non-linear codebook kvalues_iq4nl
// (vec_dot_iq4_nl_q8_0, src/ggml-cpu/ggml-cpu-quants.c:10370-10561): Q8_0 activations, d_w d_y sum(kvalues[code] q_y)
template <int NB> struct Unit<CDNA4_IQ4_NL, NB> {
    static constexpr int UK = 32;
    struct W { float d; uint32_t wl[4], wh[4]; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const uint8_t *blk = wrow + (int64_t)u * 18;
        W r; r.d = h2f(ld_u16(blk));
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t q = ld_u32_a2(blk + 2 + 4 * i); r.wl[i] = iq4nl_lut4(q & 0x0F0F0F0Fu); r.wh[i] = iq4nl_lut4((q >> 4) & 0x0F0F0F0Fu); }
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = u * 32;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16);
            const uint32_t yl[4] = {y0.x, y0.y, y0.z, y0.w}, yh[4] = {y1.x, y1.y, y1.z, y1.w};
            int s = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { s = dot4(wr.wl[i], yl[i], s); s = dot4(wr.wh[i], yh[i], s); }
            acc[c] += (wr.d * a.d[(int64_t)col[c] * (a.K / 32) + u]) * (float)s;
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- IQ4_XS: 136-byte superblock {fp16 d, u16 scales_h, scales_l[4], qs[128]}: eight 32-weight sub-blocks laid out like IQ4_NL blocks, 6-bit
// ---- scale ls of sub-block ib = (scales_l[ib / 2] >> 4 (ib % 2)) & 15 | ((scales_h >> 2 ib) & 3) << 4, weight = d (ls - 32) kvalues[code]
// ---- (vec_dot_iq4_xs_q8_K, src/ggml-cpu/ggml-cpu-quants.c:10563-10898: Q8_K activations, per sub-block (d d_y) (ls - 32) * sum).
// ---- Unit = 64 consecutive k = sub-blocks 2g, 2g+1 of superblock sb (32 bytes of qs)
template <int NB> struct Unit<CDNA4_IQ4_XS, NB> {
    static constexpr int UK = 64;
    struct W { float d; int ls0, ls1; uint32_t w[4][4]; };      // w[2 s + h][i]: sub-block 2g+s, low (h = 0: k 0..15) / high (k 16..31) codes as int8 codebook values
    __device__ static W load(const uint8_t *wrow, int u) {
        const int sb = u >> 2, g = u & 3;
        const uint8_t *blk = wrow + (int64_t)sb * 136;
        W r; r.d = h2f(ld_u16(blk));
        const uint32_t sh = ld_u16(blk + 2), sl = blk[4 + g];
        r.ls0 = (int)((sl & 0xFu) | (((sh >> (4 * g)) & 3u) << 4)) - 32;
        r.ls1 = (int)((sl >> 4) | (((sh >> (4 * g + 2)) & 3u) << 4)) - 32;
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t q = ld_u32_a2(blk + 8 + 32 * g + 16 * s + 4 * i);
                r.w[2 * s][i] = iq4nl_lut4(q & 0x0F0F0F0Fu); r.w[2 * s + 1][i] = iq4nl_lut4((q >> 4) & 0x0F0F0F0Fu);
            }
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, g = u & 3;
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = sb * 256 + 64 * g;
            const u32x4 y[4] = {act_ld16(a, yr, yo), act_ld16(a, yr, yo + 16), act_ld16(a, yr, yo + 32), act_ld16(a, yr, yo + 48)};
            int s0 = 0, s1 = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t ya[4] = {y[h].x, y[h].y, y[h].z, y[h].w}, yb[4] = {y[2 + h].x, y[2 + h].y, y[2 + h].z, y[2 + h].w};
#pragma unroll
                for (int i = 0; i < 4; i++) { s0 = dot4(wr.w[h][i], ya[i], s0); s1 = dot4(wr.w[2 + h][i], yb[i], s1); }
            }
            const float d4d8 = wr.d * a.d[(int64_t)col[c] * (a.K / 256) + sb];
            acc[c] += (d4d8 * (float)wr.ls0) * (float)s0;
            acc[c] += (d4d8 * (float)wr.ls1) * (float)s1;
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- Q2_K: 84-byte superblock {scales[16] (4-bit scale | 4-bit min << 4), qs[64], fp16 d, dmin}.  Unit = 64 consecutive k =
// ---- (128-half n, shift pair j): the 32 bytes qs[32n..] at shifts 4j and 4j+2, sub-block scales scales[8n + 4j .. + 3]
template <int NB> struct Unit<CDNA4_Q2_K, NB> {
    static constexpr int UK = 64;
    struct W { uint32_t q[8]; uint32_t sc; float d, dmin; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const int sb = u >> 2, n = (u >> 1) & 1, j = u & 1;
        const uint8_t *blk = wrow + (int64_t)sb * 84;
        W r; r.sc = ld_u32_a2(blk + 8 * n + 4 * j); r.d = h2f(ld_u16(blk + 80)); r.dmin = h2f(ld_u16(blk + 82));
#pragma unroll
        for (int i = 0; i < 8; i++) r.q[i] = ld_u32_a2(blk + 16 + 32 * n + 4 * i);
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, idx = u & 3, j = u & 1;
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = sb * 256 + 64 * idx;
            const int16_t *bs = a.bsums + (int64_t)col[c] * (a.K / 16) + sb * 16 + 4 * idx;
            int isum = 0, summs = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {                                   // sub-block t = (shift 2j + (t >> 1), byte half t & 1): k 16 t .. 16 t + 15 of the unit
                const int sh = 2 * (2 * j + (t >> 1)), half = t & 1;
                const u32x4 yv = act_ld16(a, yr, yo + 16 * t);
                const uint32_t yy[4] = {yv.x, yv.y, yv.z, yv.w};
                int s = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) s = dot4((wr.q[4 * half + i] >> sh) & 0x03030303u, yy[i], s);
                const int scb = (int)((wr.sc >> (8 * t)) & 0xFF);
                isum += (scb & 0xF) * s; summs += (scb >> 4) * (int)bs[t];
            }
            const float yd = a.d[(int64_t)col[c] * (a.K / 256) + sb];
            acc[c] += (yd * wr.d) * (float)isum - (yd * wr.dmin) * (float)summs;
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- Q3_K: 110-byte superblock {hmask[32], qs[64], scales[12] (sixteen 6-bit, -32), fp16 d}: value = (2 bits | hmask bit << 2) - 4.
// ---- Same unit as Q2_K; the unit's four scales are the byte lanes of (low nibbles of scales[4j..4j+3], high for n = 1) |
// ---- ((scales[8..11] >> 2 (2n + j)) & 3) << 4
template <int NB> struct Unit<CDNA4_Q3_K, NB> {
    static constexpr int UK = 64;
    struct W { uint32_t q[8], hm[8]; uint32_t sc; float d; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const int sb = u >> 2, n = (u >> 1) & 1, j = u & 1;
        const uint8_t *blk = wrow + (int64_t)sb * 110;                       // 2-byte aligned only
        W r; r.d = h2f(ld_u16(blk + 108));
        const uint32_t lo = ld_u32_a2(blk + 96 + 4 * j), hi = ld_u32_a2(blk + 104);
        r.sc = ((n ? (lo >> 4) : lo) & 0x0F0F0F0Fu) | (((hi >> (2 * (2 * n + j))) & 0x03030303u) << 4);
#pragma unroll
        for (int i = 0; i < 8; i++) { r.q[i] = ld_u32_a2(blk + 32 + 32 * n + 4 * i); r.hm[i] = ld_u32_a2(blk + 4 * i); }
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, idx = u & 3, n = (u >> 1) & 1, j = u & 1;
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = sb * 256 + 64 * idx;
            const int16_t *bs = a.bsums + (int64_t)col[c] * (a.K / 16) + sb * 16 + 4 * idx;
            int isum = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int s2 = 2 * j + (t >> 1), half = t & 1;               // shift index 0..3 within the 128-half; hmask bit 4n + s2
                const u32x4 yv = act_ld16(a, yr, yo + 16 * t);
                const uint32_t yy[4] = {yv.x, yv.y, yv.z, yv.w};
                int s = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t v = ((wr.q[4 * half + i] >> (2 * s2)) & 0x03030303u) | (((wr.hm[4 * half + i] >> (4 * n + s2)) & 0x01010101u) << 2);
                    s = dot4(v, yy[i], s);
                }
                isum += ((int)((wr.sc >> (8 * t)) & 0xFF) - 32) * (s - 4 * (int)bs[t]);     // sum (v - 4) y = sum v y - 4 bsum
            }
            acc[c] += (wr.d * a.d[(int64_t)col[c] * (a.K / 256) + sb]) * (float)isum;
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- Q4_0: unit = 18-byte block, nibble j -> k j (low), j+16 (high), value q-8 ------------------------------
template <int NB> struct Unit<CDNA4_Q4_0, NB> {
    static constexpr int UK = 32;
    struct W { float d; uint32_t w[4]; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const uint8_t *blk = wrow + (int64_t)u * 18;
        W r; r.d = h2f(ld_u16(blk));
#pragma unroll
        for (int i = 0; i < 4; i++) r.w[i] = ld_u32_a2(blk + 2 + 4 * i);
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = u * 32;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16);
            const uint32_t yl[4] = {y0.x, y0.y, y0.z, y0.w}, yh[4] = {y1.x, y1.y, y1.z, y1.w};
            int s = 0, ys = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s = dot4(wr.w[i] & 0x0F0F0F0Fu, yl[i], s); s = dot4((wr.w[i] >> 4) & 0x0F0F0F0Fu, yh[i], s);
                ys = dot4(0x01010101u, yl[i], ys); ys = dot4(0x01010101u, yh[i], ys);
            }
            acc[c] += (float)(s - 8 * ys) * wr.d * a.d[(int64_t)col[c] * (a.K / 32) + u];
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

// ---- Q8_0: unit = 34-byte block ----------------------------------------------------------------------------
template <int NB> struct Unit<CDNA4_Q8_0, NB> {
    static constexpr int UK = 32;
    struct W { float d; uint32_t w[8]; };
    __device__ static W load(const uint8_t *wrow, int u) {
        const uint8_t *blk = wrow + (int64_t)u * 34;
        W r; r.d = h2f(ld_u16(blk));
#pragma unroll
        for (int i = 0; i < 8; i++) r.w[i] = ld_u32_a2(blk + 2 + 4 * i);
        return r;
    }
    template <typename ACT> __device__ static void mac(const W &wr, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *yr = a.qs + (int64_t)col[c] * a.K; const int yo = u * 32;
            const u32x4 y0 = act_ld16(a, yr, yo), y1 = act_ld16(a, yr, yo + 16);
            const uint32_t yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            int s = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) s = dot4(wr.w[i], yv[i], s);
            acc[c] += (float)s * (wr.d * a.d[(int64_t)col[c] * (a.K / 32) + u]);
        }
    }
    template <typename ACT> __device__ static void dot(const uint8_t *wrow, int u, const ACT &a, const int (&col)[NB], float (&acc)[NB]) { mac(load(wrow, u), u, a, col, acc); }
};

template <int TYPE, int NB, bool IDS, int ROWS>
__global__ __launch_bounds__(256) void k_gemv_q(const cdna4_gemv_args a) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= a.M) return;                                                 // wave-uniform
    const uint8_t *Wb = a.W;
    int col[NB], ycol[NB];
    if (IDS) {
        const int c = blockIdx.y, t = c / a.n_used, u = c % a.n_used;
        const int e = a.ids[(int64_t)t * a.ids_tok_stride + u];
        if (e < 0 || e >= a.n_expert) return;
        Wb += (int64_t)e * a.w_expert_bytes;
        col[0] = t * a.n_b + (u % a.n_b); ycol[0] = c;
    } else {
#pragma unroll
        for (int c = 0; c < NB; c++) { col[c] = min((int)blockIdx.y * NB + c, a.ncol - 1); ycol[c] = blockIdx.y * NB + c; }      // (columns past ncol: padding of the 1 / 4 / 8-column forms — they repeat the last one and are not stored)
    }
    // ROWS weight rows per wave: the activation loads (same addresses for every row) are issued once and the
    // rows' superblock loads are all in flight together
    float acc[ROWS][NB];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int c = 0; c < NB; c++) acc[r][c] = 0.f;
    const int nunits = a.K / Unit<TYPE, NB>::UK;
    for (int u = lane; u < nunits; u += 64) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int row = min(row0 + r, a.M - 1);
            Unit<TYPE, NB>::dot(Wb + (int64_t)row * a.w_row_bytes, u, a, col, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const float s = wave_sum_dpp(acc[r][c]);
            if (lane == 0 && row0 + r < a.M && (IDS || ycol[c] < a.ncol)) a.Y[(int64_t)ycol[c] * a.y_col_stride + row0 + r] = epilogue_apply(a.epi, s, row0 + r, ycol[c]);
        }
}

template <int TYPE, int NB>
static void launch_nb(const cdna4_gemv_args &a, hipStream_t st) {
    // one row per wave.  (2 and 4 rows per wave — fewer, fatter waves sharing the activation loads — measured SLOWER on
    // MI355X: 6.7 vs 5.3 us cold at 4096x4096.)
    hipLaunchKernelGGL((k_gemv_q<TYPE, NB, false, 1>), dim3((a.M + 3) / 4, 1), dim3(256), 0, st, a);
}
template <int TYPE>
static int launch_type(const cdna4_gemv_args &a0, hipStream_t st) {
    cdna4_gemv_args a = a0;
    if (a.ids) {
        hipLaunchKernelGGL((k_gemv_q<TYPE, 1, true, 1>), dim3((a.M + 3) / 4, a.ncol), dim3(256), 0, st, a);
        CDNA4_CHECK_LAUNCH();
        return 0;
    }
    // groups of up to 8 columns share one pass over the weights
    for (int c0 = 0; c0 < a0.ncol; c0 += 8) {
        const int nb = a0.ncol - c0 < 8 ? a0.ncol - c0 : 8;
        a.qs = a0.qs + (int64_t)c0 * a0.K;
        a.d = a0.d + (int64_t)c0 * (a0.K / (QT<TYPE>::KQ ? 256 : 32));
        a.bsums = a0.bsums ? a0.bsums + (int64_t)c0 * (a0.K / 16) : nullptr;
        a.Y = a0.Y + (int64_t)c0 * a0.y_col_stride;
        // two column counts are instantiated (round 4 had eight, round 5 first three — 24 .. 60 kernels of a fall-back path that AUTO reaches only where neither the
        // one-launch nor the staged form exists): a group of 2..8 runs the 8-column form, the padding columns repeat the group's last one and are not stored (the
        // same columns meet the same sums: results unchanged)
        a.ncol = nb;
        if (nb == 1) launch_nb<TYPE, 1>(a, st); else launch_nb<TYPE, 8>(a, st);
        CDNA4_CHECK_LAUNCH();
    }
    return 0;
}



// ---- fused activation-quantize + GEMV (single-column decode) ------------------------------------------------
// One launch instead of two for the B=1 MUL_MAT: every workgroup re-quantizes the whole activation row into
// LDS (K bytes of int8 + scales + bsums; the row is 16 KiB at K=4096 and L2-resident after the first
// workgroup touched it), then its four waves run the same Unit::dot bodies against LDS.  The redundant
// quantization costs ~0.5 us of VALU per workgroup and saves a ~3 us kernel plus the launch gap.
// (Touching the wave's weight row before the quantization to overlap its HBM latency was measured SLOWER: 7.2 vs 6.1 us.)

// NW waves per work-group, ROWS weight rows per wave.  The quantizer's cost is per WORK-GROUP (every work-group redoes the
// whole row), so fewer, fatter work-groups pay it less often: <4,1> = 1024 work-groups at M=4096, <8,2> = 256 (one per CU).
// IDS: single-token MUL_MAT_ID (mixture-of-experts decode): blockIdx.y = slot u, expert = ids[u] read on the device, the
// slot's activation row x + (u % n_b) * x_row_stride, output column u.
// NB > 1: the same for up to NB activation rows at once (small-batch decode, a.ncol <= NB rows x_row_stride apart): the work-group
// quantizes all of them into LDS and every weight unit, loaded once, meets NB columns that are read from LDS — the two-launch
// k_gemv_q<.., NB> reads its activations from global memory again for every weight row (512 B of L1 traffic per 36-B weight unit
// at NB = 8: 43 us at 4096 x 14336 where one pass over the weights takes 10).
// PREQ: the activation rows arrive already quantized (a.qs / a.d / a.bsums, written once by k_quantize_q8_K / q8_0) and are only COPIED
// into LDS — for NB x K beyond ~32 K values the redundant per-work-group quantization (and its 4 K bytes of fp32 reads per value row and
// work-group) costs more than the extra launch: measured 40 us at 8 x 14336 with the quantizer inside.
// (Round 4's DMA form — whole weight rows requested by LDS-DMA up front — was a measured loss and went in round 5: profiles/r04/decode_ab.txt.)
// (the body as a device function: k_gemv_q_fused below, and k_gemv_q_fused_grp — several weight matrices against ONE activation row in one launch)
extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
template <int TYPE, int NW, int ROWS, bool IDS = false, int NB = 1, bool PREQ = false>
__device__ __forceinline__ void gemv_q_fused_body(cdna4_gemv_args a, const float *__restrict__ x, int64_t x_row_stride) {
    constexpr bool KQ = QT<TYPE>::KQ;
    if (IDS) {
        const int u = blockIdx.y, e = a.ids[u];
        if (e < 0 || e >= a.n_expert) return;                              // whole work-group
        a.W += (int64_t)e * a.w_expert_bytes; x += (int64_t)(u % a.n_b) * x_row_stride; a.Y += (int64_t)u * a.y_col_stride;
    }
    const int K = a.K, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = K / 16, nqd = K / (KQ ? 256 : 32);                    // 16-value chunks / scales per activation row
    int8_t *sq = reinterpret_cast<int8_t *>(smem);                          // [NB][K]
    constexpr bool Q81 = cdna4_is_q81(TYPE);                                // Q8_1 activations: fp32 s[NB][K/32] in the place of the bsums
    int16_t *sbs = reinterpret_cast<int16_t *>(smem + NB * K);              // [NB][K/16] (Q8_K only)
    float *sd = reinterpret_cast<float *>(smem + NB * K + ((KQ || Q81) ? NB * (K / 8) : 0));   // [NB][nqd]
    const int row0 = (blockIdx.x * NW + wave) * ROWS;
    const int nunits = K / Unit<TYPE, NB>::UK;
    const int total = NB * nch;                                             // chunk id = col * nch + c; 16 adjacent ids share a column and a superblock
    auto xrow = [&](int col) -> const float * { return x + (int64_t)(NB == 1 ? 0 : min(col, a.ncol - 1)) * x_row_stride; };   // (padding columns repeat the last row)
    // Order matters: a wave's loads RETURN in issue order.  The activation chunk (L2-resident) is requested first so the
    // quantizer can start as soon as it arrives; the weight rows (HBM) are requested right behind it and stream in under the
    // quantizer.  (Weights first made the quantizer wait for the whole HBM round trip: 5.46 vs 4.82 us cold, same-box A/B.)
    const int c_first = threadIdx.x;
    float4 v_first[4] = {};
    const uint8_t *wrow[ROWS];
    typename Unit<TYPE, NB>::W w0[ROWS];
    if (!PREQ && c_first < total) {
        const float *px = xrow(c_first / nch) + (c_first % nch) * 16;
#pragma unroll
        for (int i = 0; i < 4; i++) v_first[i] = *reinterpret_cast<const float4 *>(px + 4 * i);
    }
    asm volatile("" ::: "memory");                                          // keep the weight loads below behind the activation loads
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        wrow[r] = a.W + (int64_t)min(row0 + r, a.M - 1) * a.w_row_bytes;
        w0[r] = Unit<TYPE, NB>::load(wrow[r], min(lane, nunits - 1));
    }
    // One lane quantizes 16 consecutive values (= one bsums entry, one ds_write_b128): a superblock is 16 adjacent lanes
    // (4 butterfly rounds), a Q8_0 block 2 lanes (1 round).  The first cut (one wave per superblock, 4 values per lane,
    // 6 rounds x 3 shuffles, 4 superblocks in sequence per wave) cost ~5 us per work-group and lost to the two-kernel path.
    auto quantize_chunk = [&](int id, const float4 (&v)[4]) __attribute__((always_inline)) {
        const int col = NB == 1 ? 0 : id / nch, c = NB == 1 ? id : id % nch;
        const float e[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
        int q[16];
        if (KQ) {
            // first index with the largest |x| keeps its SIGNED value (quantize_row_q8_K_ref, src/ggml-quants.c:2485-2491)
            float amax = 0.f, mx = 0.f; int idx = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) { const float ax = fabsf(e[i]); if (ax > amax) { amax = ax; mx = e[i]; idx = (c & 15) * 16 + i; } }
            // 16-lane all-reduce on the VALU (cdna4_common.h: dpp_*): the selection (largest |x|, then smallest index) is commutative and associative
            auto take = [&](float oa, float om, int oi) __attribute__((always_inline)) { if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; } };
            take(dpp_f32<0xB1>(amax), dpp_f32<0xB1>(mx), dpp_i32<0xB1>(idx));
            take(dpp_f32<0x4E>(amax), dpp_f32<0x4E>(mx), dpp_i32<0x4E>(idx));
            take(dpp_f32<0x141>(amax), dpp_f32<0x141>(mx), dpp_i32<0x141>(idx));
            take(dpp_f32<0x140>(amax), dpp_f32<0x140>(mx), dpp_i32<0x140>(idx));
            float d = 0.f; int bsum = 0;
            if (amax != 0.f) {
                const float iscale = -127.f / mx;
#pragma unroll
                for (int i = 0; i < 16; i++) { const int t = (int)__builtin_rintf(iscale * e[i]); q[i] = t < 127 ? t : 127; bsum += q[i]; }
                d = 1.0f / iscale;
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) q[i] = 0;
            }
            sbs[col * nch + c] = (int16_t)bsum;
            if ((c & 15) == 0) sd[col * nqd + (c >> 4)] = d;
        } else {
            // AVX2 body of quantize_row_q8_0 (src/ggml-cpu/ggml-cpu-quants.c:778-815): d = amax/127 -> fp16, id = 127/amax, RNE
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) amax = fmaxf(amax, fabsf(e[i]));
            amax = fmaxf(amax, dpp_f32<0xB1>(amax));                        // the block's other 16 values: the neighbouring lane
            const float d = amax / 127.f, id = amax != 0.f ? 127.f / amax : 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) q[i] = (int)__builtin_rintf(e[i] * id);
            if ((c & 1) == 0) sd[col * nqd + (c >> 1)] = h2f(f2h_bits(d));
            if constexpr (Q81) {
                // quantize_row_q8_1 (AVX2 body, src/ggml-cpu/ggml-cpu-quants.c:1076-1119): s = fp16(d * sum of the 32 quants), d still in fp32
                int sum = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) sum += q[i];
                sum += dpp_i32<0xB1>(sum);
                if ((c & 1) == 0) reinterpret_cast<float *>(sbs)[col * nqd + (c >> 1)] = h2f(f2h_bits(d * (float)sum));
            }
        }
        u32x4 pk;
        { const int q0[4] = {q[0], q[1], q[2], q[3]}, q1[4] = {q[4], q[5], q[6], q[7]}, q2[4] = {q[8], q[9], q[10], q[11]}, q3[4] = {q[12], q[13], q[14], q[15]};
          pk.x = pack4i8(q0); pk.y = pack4i8(q1); pk.z = pack4i8(q2); pk.w = pack4i8(q3); }
        *reinterpret_cast<u32x4 *>(sq + col * K + lds_swz(c * 16)) = pk;
    };
    if constexpr (PREQ) {
        // 16-byte copies: int8 rows (K bytes each), then bsums (K/8 bytes each), then scales (4 nqd bytes each; nqd % 4 == 0 is required)
        for (int id = threadIdx.x; id < total; id += NW * 64) {
            const int col = id / nch, c = id % nch, sc = min(col, a.ncol - 1);
            *reinterpret_cast<u32x4 *>(sq + col * K + lds_swz(c * 16)) = *reinterpret_cast<const u32x4 *>(a.qs + (int64_t)sc * K + c * 16);
        }
        if (KQ) for (int id = threadIdx.x; id < NB * (nch / 8); id += NW * 64) {
            const int col = id / (nch / 8), c = id % (nch / 8), sc = min(col, a.ncol - 1);
            *reinterpret_cast<u32x4 *>(sbs + col * nch + c * 8) = *reinterpret_cast<const u32x4 *>(a.bsums + (int64_t)sc * nch + c * 8);
        }
        for (int id = threadIdx.x; id < NB * nqd; id += NW * 64) { const int col = id / nqd, c = id % nqd; sd[col * nqd + c] = a.d[(int64_t)min(col, a.ncol - 1) * nqd + c]; }
    } else {
    if (c_first < total) quantize_chunk(c_first, v_first);
    for (int id = c_first + NW * 64; id < total; id += NW * 64) {          // more chunks than threads: the remaining ones
        const float *px = xrow(id / nch) + (id % nch) * 16;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const float4 *>(px + 4 * i);
        quantize_chunk(id, v);
    }
    }
    __syncthreads();
    if (row0 >= a.M) return;
    const lds_act act{sq, sd, sbs, K};
    int col[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) col[c] = c;
    // Rows longer than one round of 64 units (K > 4096 for the K-quants): the NEXT round's weight bytes of all ROWS rows are requested before the
    // current round's integer dots, so two rounds of the stream are in flight instead of one load -> use chain per round (4096 x 14336: 12.8 us
    // for a 33 MB stream whose floor is 5.3 — VERDICT r2 weak 6).  Per row the units are still added in the order u = lane, lane + 64, ...:
    // bit-identical to the one-round-at-a-time loop.
    float acc[ROWS][NB];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int c = 0; c < NB; c++) acc[r][c] = 0.f;
    typename Unit<TYPE, NB>::W cur[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) cur[r] = w0[r];
    constexpr bool PIPE = NB <= 2;                                          // (4 and 8 columns: the second set of weight registers spills, and those forms are bound by their integer dots)
    // SWAP: two rounds per trip with the two register sets trading places, where that does not cost occupancy: with one row per wave in 4- / 8-wave work-groups (several per
    // CU on tall matrices) the swapped form takes 81 instead of 61-64 registers for Q4_K and 14336 x 4096 went from 9.5 to 11.0 us (round 5, one box); with two rows per wave it
    // takes FEWER (115 vs 128) and 4096 x 14336 on 8 x 2 went from 11.2 to 10.3 us; 16-wave work-groups run one per CU, four waves per SIMD, whatever the count
    constexpr bool SWAP = PIPE && (ROWS == 2 || NW == 16);
    if constexpr (SWAP) {
        // two rounds per trip with the two register sets trading places (round 4 copied `nxt` into `cur` every round: 13 register moves per row and round in an
        // instruction-bound kernel, profiles/r04/pmc_decode_counters.txt); the units of a row are still added in the order u = lane, lane + 64, ..
        typename Unit<TYPE, NB>::W alt[ROWS];
        for (int u = lane; u < nunits; u += 128) {
            const int u1 = u + 64, u2 = u + 128;
            if (u1 < nunits) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) alt[r] = Unit<TYPE, NB>::load(wrow[r], u1);
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) Unit<TYPE, NB>::mac(cur[r], u, act, col, acc[r]);
            if (u1 < nunits) {
                if (u2 < nunits) {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) cur[r] = Unit<TYPE, NB>::load(wrow[r], u2);
                }
#pragma unroll
                for (int r = 0; r < ROWS; r++) Unit<TYPE, NB>::mac(alt[r], u1, act, col, acc[r]);
            }
        }
    } else if constexpr (PIPE) {
        for (int u = lane; u < nunits; u += 64) {
            typename Unit<TYPE, NB>::W nxt[ROWS];
            const int un = u + 64;
            if (un < nunits) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) nxt[r] = Unit<TYPE, NB>::load(wrow[r], un);
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) Unit<TYPE, NB>::mac(cur[r], u, act, col, acc[r]);
            if (un < nunits) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) cur[r] = nxt[r];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (lane < nunits) Unit<TYPE, NB>::mac(cur[r], lane, act, col, acc[r]);
            for (int u = lane + 64; u < nunits; u += 64) Unit<TYPE, NB>::dot(wrow[r], u, act, col, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const float s = wave_sum_dpp(acc[r][c]);
            if (lane == 0 && row0 + r < a.M && c < (NB == 1 ? 1 : a.ncol)) a.Y[(int64_t)c * a.y_col_stride + row0 + r] = epilogue_apply(a.epi, s, row0 + r, c);
        }
}

template <int TYPE, int NW, int ROWS, bool IDS = false, int NB = 1, bool PREQ = false>
__global__ __launch_bounds__(NW * 64) void k_gemv_q_fused(cdna4_gemv_args a, const float *__restrict__ x, int64_t x_row_stride) {
    gemv_q_fused_body<TYPE, NW, ROWS, IDS, NB, PREQ>(a, x, x_row_stride);
}
// Several MUL_MATs of ONE activation row in one launch (round 6, VERDICT r5 item 5: wq / wk / wv and w_gate / w_up of a decoded token share src1 — three launch ramps and
// three kernel boundaries where one does): blockIdx.y = the matrix, blockIdx.x = its block of NW * ROWS rows (blocks past a matrix's rows leave at once).  Every work-group
// runs the body of k_gemv_q_fused on its matrix: the same quantizer, the same dots in the same order — bit-identical to the separate calls.
template <int TYPE, int NW, int ROWS>
__global__ __launch_bounds__(NW * 64) void k_gemv_q_fused_grp(const cdna4_gemv_group g, const float *__restrict__ x) {
    const int i = blockIdx.y;
    if ((int)(blockIdx.x * NW * ROWS) >= g.M[i]) return;                  // (the whole work-group, before any barrier)
    cdna4_gemv_args a{};
    a.type = TYPE; a.W = g.W[i]; a.w_row_bytes = g.w_row_bytes[i]; a.Y = g.Y[i]; a.y_col_stride = g.M[i]; a.M = g.M[i]; a.K = g.K; a.ncol = 1; a.ids = nullptr;
    a.epi.bias = g.bias[i];
    gemv_q_fused_body<TYPE, NW, ROWS>(a, x, 0);
}

size_t cdna4_gemv_fused_lds_bytes(int type, int64_t K) {
    const bool kq = type == CDNA4_Q4_K || type == CDNA4_Q5_K || type == CDNA4_Q6_K || type == CDNA4_Q2_K || type == CDNA4_Q3_K || type == CDNA4_IQ4_XS;
    return (size_t)(K + (kq ? K / 8 + (K / 256) * 4 : (K / 32) * 4) + (cdna4_is_q81(type) ? K / 8 : 0));
}
static int fused_nb(int64_t B) { return B <= 1 ? 1 : (B <= 2 ? 2 : (B <= 4 ? 4 : 8)); }      // instantiated column counts
bool cdna4_gemv_fused_supported(int type, int64_t K, int64_t B) {
    if (K <= 0 || B < 1 || B > 8) return false;
    // lds_swz() permutes the 16-byte chunks of an activation row inside groups of four (chunk c -> c ^ ((c >> 4) & 3)): a row that is not a
    // whole number of 64-byte groups (32-weight formats with K % 64 == 32) would have its last two chunks land PAST the row for K mod 1024 >=
    // 512 — on the row's scales.  Found by the CPU emulation of the whole library (K = 544); such K take the quantize + GEMV pair instead.
    if (K % 64) return false;
    if (B == 1) return cdna4_gemv_fused_lds_bytes(type, K) <= 64 * 1024;
    const bool main5 = type == CDNA4_Q4_K || type == CDNA4_Q5_K || type == CDNA4_Q6_K || type == CDNA4_Q4_0 || type == CDNA4_Q8_0;
    return main5 && (size_t)fused_nb(B) * cdna4_gemv_fused_lds_bytes(type, K) <= 150 * 1024;
}
template <int TYPE>
static int launch_fused(const cdna4_gemv_args &a, const float *x, hipStream_t st) {
    const size_t lds = cdna4_gemv_fused_lds_bytes(TYPE, a.K);
    // CDNA4_FUSED_CFG: 0 = 4 waves x 1 row, 1 = 8 x 2, 2 = 4 x 2, 3 = 8 x 1.  Measured at 4096x4096 Q4_K with the activation loads
    // issued first, cold HBM / host wall us: 5.80/5.22, 4.35/4.29, 5.38/4.76, 4.83/4.45 -> 8 waves x 2 rows (one work-group per
    // CU at M = 4096: the quantizer is paid once per CU) when that still gives every CU a work-group, else 8 x 1, else 4 x 1.
    static const int cfg_env = getenv("CDNA4_FUSED_CFG") ? atoi(getenv("CDNA4_FUSED_CFG")) : -1;
    // Round 4 (profiles/r04/decode_cfg.txt, one box, us cold, 8 x 2 vs 8 x 1): rows longer than one round of 64 units stream better from twice as many, half as fat
    // work-groups — Q4_K 4096 x 14336 11.22 vs 10.64, 4096 x 11008 9.41 vs 8.49, 4096 x 8192 7.41 vs 6.68, 14336 x 4096 10.31 vs 9.45 (4096^2: 4.24 vs 4.56, 11008 x 4096 8.10
    // vs 8.25: stay); Q6_K 14336 x 4096 17.67 vs 14.48, 11008 x 4096 13.88 vs 12.17 (its K-long shapes: level); Q4_0: 8 x 2 everywhere.
    int cfg = cfg_env >= 0 ? cfg_env : (a.M >= 4096 ? 1 : (a.M >= 2048 ? 3 : 0));
    if (cfg_env < 0 && cfg == 1) {
        if (TYPE == CDNA4_Q4_K && (a.K >= 8192 || a.M >= 12288)) cfg = 3;
        if (TYPE == CDNA4_Q6_K && a.M >= 8192) cfg = 3;
    }
    // Round 5: 16 waves x 1 row (cfg 4: 1024-thread work-groups, the quantizer still paid once per 16 rows like 8 x 2, but every wave has ONE row to multiply and the
    // activation row is quantized by twice as many lanes) wherever that is one work-group per CU at most.  One box, us cold, the launcher's earlier choice vs 16 x 1
    // (profiles/r05/decode_cfg.txt): Q4_K 4096 x 14336 10.67 vs 9.65, 4096 x 11008 8.78 vs 8.01, 4096 x 8192 6.90 vs 6.46, 4096^2 4.17 vs 4.12; Q5_K 12.62 vs 11.30, 9.91 vs 9.17,
    // 8.01 vs 7.26, 4.78 vs 4.28; Q6_K 15.09 vs 14.98, 11.83 vs 11.45, 9.21 vs 8.54, 5.67 vs 5.65.  Taller matrices (several work-groups per CU) keep the rules above.
    // Q4_0: 4.55 vs 4.42, 10.81 vs 10.60, 9.34 vs 9.03, 7.11 vs 6.84 (same order of shapes); Q8_0 keeps 8 x 2 (long rows lose: 16.6 vs 17.3 at 4096 x 14336).
    if (cfg_env < 0 && (QT<TYPE>::KQ || TYPE == CDNA4_Q4_0) && (a.M + 15) / 16 <= cdna4_gemm_cu_count() && a.M >= 2048) cfg = 4;
    // (the LDS-DMA form of the 8 x 2 configuration — whole weight rows requested up front, CDNA4_DECODE_DMA — was a measured loss in round 4 (profiles/r04/decode_ab.txt: 4096 x 14336
    //  11.19 vs 11.89 us) and was removed in round 5; git show 5eb5f7c:ggml_amd/csrc/gemv_q.hip)
    constexpr bool HAS16 = QT<TYPE>::KQ || TYPE == CDNA4_Q4_0;          // (the formats whose rule can choose 16 x 1; the others keep 8 x 2 behind CDNA4_FUSED_CFG=4)
    if (cfg == 2) cfg = 1;                                              // (4 waves x 2 rows: never AUTO's choice; its instantiations went in round 5)
    if (cfg == 4 && !HAS16) cfg = 1;
    if constexpr (HAS16) { if (cfg == 4) { hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 16, 1>), dim3((a.M + 15) / 16), dim3(1024), lds, st, a, x, (int64_t)0); CDNA4_CHECK_LAUNCH(); return 0; } }
    if (cfg == 1) hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 8, 2>), dim3((a.M + 15) / 16), dim3(512), lds, st, a, x, (int64_t)0);
    else if (cfg == 3) hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 8, 1>), dim3((a.M + 7) / 8), dim3(512), lds, st, a, x, (int64_t)0);
    else hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 4, 1>), dim3((a.M + 3) / 4), dim3(256), lds, st, a, x, (int64_t)0);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
// 2..8 activation rows: 8 waves x 2 rows per work-group; more than 64 KB of LDS needs the attribute once per kernel
template <int TYPE, int NB>
static int launch_fused_nb(const cdna4_gemv_args &a, const float *x, int64_t x_row_stride, hipStream_t st) {
    const size_t lds = (size_t)NB * cdna4_gemv_fused_lds_bytes(TYPE, a.K);
    static bool raised_[16] = {};                                       // (a function attribute is per device)
    int dev_ = 0; if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 16) { (void)hipGetLastError(); dev_ = 0; }
    bool &raised = raised_[dev_];
    if (!raised) {
        if (hipFuncSetAttribute((const void *)k_gemv_q_fused<TYPE, 8, 2, false, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { (void)hipGetLastError(); return cdna4_set_error_msg("gemv_q_fused: cannot raise the dynamic LDS limit"); }
        if constexpr (NB <= 4 && (QT<TYPE>::KQ || TYPE == CDNA4_Q4_0) && !(TYPE == CDNA4_Q5_K && NB == 2)) {      // (Q5_K x 2 columns needs more than the 128 registers of a 16-wave work-group)
            if (hipFuncSetAttribute((const void *)k_gemv_q_fused<TYPE, 16, 1, false, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { (void)hipGetLastError(); return cdna4_set_error_msg("gemv_q_fused: cannot raise the dynamic LDS limit"); }
        }
        raised = true;
    }
    // Round 6: 2 and 4 rows take the 16-wave x 1-row work-groups of the one-row decode (round 5: the quantizer is still paid once per 16 weight rows, by twice as many
    // lanes, and every wave multiplies ONE row) wherever that is one work-group per CU at most — CDNA4_FUSED_NB_CFG=0 keeps 8 x 2
    static const int cfg_env = getenv("CDNA4_FUSED_NB_CFG") ? atoi(getenv("CDNA4_FUSED_NB_CFG")) : -1;
    if constexpr (NB <= 4 && (QT<TYPE>::KQ || TYPE == CDNA4_Q4_0) && !(TYPE == CDNA4_Q5_K && NB == 2)) {      // (Q5_K x 2 columns needs more than the 128 registers of a 16-wave work-group)
        if (cfg_env != 0 && (a.M + 15) / 16 <= cdna4_gemm_cu_count() && a.M >= 2048) {
            hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 16, 1, false, NB>), dim3((a.M + 15) / 16), dim3(1024), lds, st, a, x, x_row_stride);
            CDNA4_CHECK_LAUNCH();
            return 0;
        }
    }
    hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 8, 2, false, NB>), dim3((a.M + 15) / 16), dim3(512), lds, st, a, x, x_row_stride);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
template <int TYPE>
static int launch_fused_n(const cdna4_gemv_args &a, const float *x, int64_t x_row_stride, hipStream_t st) {
    switch (fused_nb(a.ncol)) { case 2: return launch_fused_nb<TYPE, 2>(a, x, x_row_stride, st); case 4: return launch_fused_nb<TYPE, 4>(a, x, x_row_stride, st); default: return launch_fused_nb<TYPE, 8>(a, x, x_row_stride, st); }
}
template <int TYPE, int NB>
static int launch_staged_nb(const cdna4_gemv_args &a, hipStream_t st) {
    const size_t lds = (size_t)NB * cdna4_gemv_fused_lds_bytes(TYPE, a.K);
    static bool raised_[16] = {};                                       // (a function attribute is per device)
    int dev_ = 0; if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 16) { (void)hipGetLastError(); dev_ = 0; }
    bool &raised = raised_[dev_];
    if (!raised) { if (hipFuncSetAttribute((const void *)k_gemv_q_fused<TYPE, 8, 2, false, NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { (void)hipGetLastError(); return cdna4_set_error_msg("gemv_q_staged: cannot raise the dynamic LDS limit"); } raised = true; }
    hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 8, 2, false, NB, true>), dim3((a.M + 15) / 16), dim3(512), lds, st, a, (const float *)nullptr, (int64_t)0);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
template <int TYPE>
static int launch_staged(const cdna4_gemv_args &a, hipStream_t st) {
    switch (fused_nb(a.ncol)) { case 2: return launch_staged_nb<TYPE, 2>(a, st); case 4: return launch_staged_nb<TYPE, 4>(a, st); default: return launch_staged_nb<TYPE, 8>(a, st); }
}
// 2..8 PRE-QUANTIZED activation rows (a.qs / a.d / a.bsums as ggml_cdna4_prepare_act lays them out), staged into LDS once per work-group
bool cdna4_gemv_staged_supported(int type, int64_t K, int64_t B) {
    const int64_t nqd = K / ((type == CDNA4_Q4_0 || type == CDNA4_Q8_0) ? 32 : 256);
    return B >= 2 && cdna4_gemv_fused_supported(type, K, B) && K % 128 == 0 && nqd > 0;
}
int cdna4_launch_gemv_q_staged(const cdna4_gemv_args &a, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (!cdna4_gemv_staged_supported(a.type, a.K, a.ncol) || a.ids) return cdna4_set_error_msg("gemv_q_staged: unsupported shape");
    if (((uintptr_t)a.qs | (uintptr_t)a.bsums) & 15) return cdna4_set_error_msg("gemv_q_staged: quantized activations must be 16-byte aligned");
    switch (a.type) {
        case CDNA4_Q4_K: case CDNA4_Q5_K:
            if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 15) return cdna4_set_error_msg("gemv_q: Q4_K/Q5_K rows must be 16-byte aligned");
            return a.type == CDNA4_Q4_K ? launch_staged<CDNA4_Q4_K>(a, st) : launch_staged<CDNA4_Q5_K>(a, st);
        case CDNA4_Q6_K: return launch_staged<CDNA4_Q6_K>(a, st);
        case CDNA4_Q4_0: return launch_staged<CDNA4_Q4_0>(a, st);
        case CDNA4_Q8_0: return launch_staged<CDNA4_Q8_0>(a, st);
    }
    return cdna4_set_error_msg("gemv_q_staged: unsupported weight type");
}
// a.ncol = 2..8 rows of x (fp32, 16-byte aligned, x_row_stride elements apart), quantized in LDS; Y column c at a.Y + c * a.y_col_stride
int cdna4_launch_gemv_q_fused_n(const cdna4_gemv_args &a, const float *x, int64_t x_row_stride, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (a.ncol < 2 || !cdna4_gemv_fused_supported(a.type, a.K, a.ncol)) return cdna4_set_error_msg("gemv_q_fused_n: unsupported shape");
    if (((uintptr_t)x | (uintptr_t)(x_row_stride * 4)) & 15) return cdna4_set_error_msg("gemv_q_fused_n: x rows must be 16-byte aligned");
    if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 1) return cdna4_set_error_msg("gemv_q: misaligned operands");
    switch (a.type) {
        case CDNA4_Q4_K: case CDNA4_Q5_K:
            if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 15) return cdna4_set_error_msg("gemv_q: Q4_K/Q5_K rows must be 16-byte aligned");
            if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256");
            return a.type == CDNA4_Q4_K ? launch_fused_n<CDNA4_Q4_K>(a, x, x_row_stride, st) : launch_fused_n<CDNA4_Q5_K>(a, x, x_row_stride, st);
        case CDNA4_Q6_K: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_fused_n<CDNA4_Q6_K>(a, x, x_row_stride, st);
        case CDNA4_Q4_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_n<CDNA4_Q4_0>(a, x, x_row_stride, st);
        case CDNA4_Q8_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_n<CDNA4_Q8_0>(a, x, x_row_stride, st);
    }
    return cdna4_set_error_msg("gemv_q_fused_n: unsupported weight type");
}

int cdna4_launch_gemv_q(const cdna4_gemv_args &a, hipStream_t st) {
    if (a.M <= 0 || a.ncol <= 0) return 0;
    if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes | (uintptr_t)a.qs) & 1) return cdna4_set_error_msg("gemv_q: misaligned operands");
    if (a.ids && a.ncol > 65535) return cdna4_set_error_msg("gemv_q: too many MUL_MAT_ID columns");
    switch (a.type) {
        case CDNA4_Q4_K: case CDNA4_Q5_K:
            if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes | (uintptr_t)a.w_expert_bytes) & 15) return cdna4_set_error_msg("gemv_q: Q4_K/Q5_K rows must be 16-byte aligned");
            if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256");
            return a.type == CDNA4_Q4_K ? launch_type<CDNA4_Q4_K>(a, st) : launch_type<CDNA4_Q5_K>(a, st);
        case CDNA4_Q6_K:
            if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256");
            return launch_type<CDNA4_Q6_K>(a, st);
        case CDNA4_Q4_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_type<CDNA4_Q4_0>(a, st);
        case CDNA4_Q8_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_type<CDNA4_Q8_0>(a, st);
        case CDNA4_Q5_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_type<CDNA4_Q5_0>(a, st);
        case CDNA4_Q4_1: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_type<CDNA4_Q4_1>(a, st);
        case CDNA4_Q5_1: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_type<CDNA4_Q5_1>(a, st);
        case CDNA4_IQ4_NL: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_type<CDNA4_IQ4_NL>(a, st);
        case CDNA4_Q2_K: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_type<CDNA4_Q2_K>(a, st);
        case CDNA4_Q3_K: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_type<CDNA4_Q3_K>(a, st);
        case CDNA4_IQ4_XS: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_type<CDNA4_IQ4_XS>(a, st);
    }
    return cdna4_set_error_msg("gemv_q: unsupported weight type");
}

// a.qs / a.d / a.bsums are ignored: the activation row x (fp32, 16-byte aligned, K values) is quantized in LDS
int cdna4_launch_gemv_q_fused(const cdna4_gemv_args &a, const float *x, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (!cdna4_gemv_fused_supported(a.type, a.K, a.ncol)) return cdna4_set_error_msg("gemv_q_fused: unsupported shape");
    if ((uintptr_t)x & 15) return cdna4_set_error_msg("gemv_q_fused: x must be 16-byte aligned");
    if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 1) return cdna4_set_error_msg("gemv_q: misaligned operands");
    switch (a.type) {
        case CDNA4_Q4_K: case CDNA4_Q5_K:
            if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 15) return cdna4_set_error_msg("gemv_q: Q4_K/Q5_K rows must be 16-byte aligned");
            if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256");
            return a.type == CDNA4_Q4_K ? launch_fused<CDNA4_Q4_K>(a, x, st) : launch_fused<CDNA4_Q5_K>(a, x, st);
        case CDNA4_Q6_K:
            if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256");
            return launch_fused<CDNA4_Q6_K>(a, x, st);
        case CDNA4_Q4_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused<CDNA4_Q4_0>(a, x, st);
        case CDNA4_Q8_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused<CDNA4_Q8_0>(a, x, st);
        case CDNA4_Q5_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused<CDNA4_Q5_0>(a, x, st);
        case CDNA4_Q4_1: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused<CDNA4_Q4_1>(a, x, st);
        case CDNA4_Q5_1: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused<CDNA4_Q5_1>(a, x, st);
        case CDNA4_IQ4_NL: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused<CDNA4_IQ4_NL>(a, x, st);
        case CDNA4_Q2_K: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_fused<CDNA4_Q2_K>(a, x, st);
        case CDNA4_Q3_K: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_fused<CDNA4_Q3_K>(a, x, st);
        case CDNA4_IQ4_XS: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_fused<CDNA4_IQ4_XS>(a, x, st);
    }
    return cdna4_set_error_msg("gemv_q: unsupported weight type");
}

// single-token MUL_MAT_ID: one launch, a.ncol = n_used slots, a.ids / w_expert_bytes / n_expert / n_b as for cdna4_launch_gemv_q
template <int TYPE>
static int launch_fused_ids(const cdna4_gemv_args &a, const float *x, int64_t x_row_stride, hipStream_t st) {
    const size_t lds = cdna4_gemv_fused_lds_bytes(TYPE, a.K);
    hipLaunchKernelGGL((k_gemv_q_fused<TYPE, 8, 1, true>), dim3((a.M + 7) / 8, a.ncol), dim3(512), lds, st, a, x, x_row_stride);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int cdna4_launch_gemv_q_fused_ids(const cdna4_gemv_args &a, const float *x, int64_t x_row_stride, hipStream_t st) {
    if (a.M <= 0 || a.ncol <= 0) return 0;
    if (!a.ids || !cdna4_gemv_fused_supported(a.type, a.K, 1) || a.ncol > 65535) return cdna4_set_error_msg("gemv_q_fused_ids: unsupported shape");
    if (((uintptr_t)x | (uintptr_t)(x_row_stride * 4)) & 15) return cdna4_set_error_msg("gemv_q_fused_ids: x rows must be 16-byte aligned");
    if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes | (uintptr_t)a.w_expert_bytes) & 1) return cdna4_set_error_msg("gemv_q: misaligned operands");
    switch (a.type) {
        case CDNA4_Q4_K: case CDNA4_Q5_K:
            if (((uintptr_t)a.W | (uintptr_t)a.w_row_bytes | (uintptr_t)a.w_expert_bytes) & 15) return cdna4_set_error_msg("gemv_q: Q4_K/Q5_K rows must be 16-byte aligned");
            if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256");
            return a.type == CDNA4_Q4_K ? launch_fused_ids<CDNA4_Q4_K>(a, x, x_row_stride, st) : launch_fused_ids<CDNA4_Q5_K>(a, x, x_row_stride, st);
        case CDNA4_Q6_K:
            if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256");
            return launch_fused_ids<CDNA4_Q6_K>(a, x, x_row_stride, st);
        case CDNA4_Q4_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_ids<CDNA4_Q4_0>(a, x, x_row_stride, st);
        case CDNA4_Q8_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_ids<CDNA4_Q8_0>(a, x, x_row_stride, st);
        case CDNA4_Q5_0: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_ids<CDNA4_Q5_0>(a, x, x_row_stride, st);
        case CDNA4_Q4_1: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_ids<CDNA4_Q4_1>(a, x, x_row_stride, st);
        case CDNA4_Q5_1: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_ids<CDNA4_Q5_1>(a, x, x_row_stride, st);
        case CDNA4_IQ4_NL: if (a.K % 32) return cdna4_set_error_msg("gemv_q: K must be a multiple of 32"); return launch_fused_ids<CDNA4_IQ4_NL>(a, x, x_row_stride, st);
        case CDNA4_Q2_K: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_fused_ids<CDNA4_Q2_K>(a, x, x_row_stride, st);
        case CDNA4_Q3_K: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_fused_ids<CDNA4_Q3_K>(a, x, x_row_stride, st);
        case CDNA4_IQ4_XS: if (a.K % 256) return cdna4_set_error_msg("gemv_q: K must be a multiple of 256"); return launch_fused_ids<CDNA4_IQ4_XS>(a, x, x_row_stride, st);
    }
    return cdna4_set_error_msg("gemv_q: unsupported weight type");
}

// n (2 .. CDNA4_GEMV_GROUP_MAX) matrices of one type and K against the SAME fp32 activation row x: one launch (k_gemv_q_fused_grp)
template <int TYPE>
static int launch_fused_grp(const cdna4_gemv_group &g, int n, const float *x, hipStream_t st) {
    const size_t lds = cdna4_gemv_fused_lds_bytes(TYPE, g.K);
    int mmax = 0; for (int i = 0; i < n; i++) mmax = g.M[i] > mmax ? g.M[i] : mmax;
    constexpr bool HAS16 = QT<TYPE>::KQ || TYPE == CDNA4_Q4_0;          // (the formats whose one-row launcher takes 16 waves x 1 row: launch_fused)
    if constexpr (HAS16) {
        if (mmax >= 2048) { hipLaunchKernelGGL((k_gemv_q_fused_grp<TYPE, 16, 1>), dim3((mmax + 15) / 16, n), dim3(1024), lds, st, g, x); CDNA4_CHECK_LAUNCH(); return 0; }
    }
    hipLaunchKernelGGL((k_gemv_q_fused_grp<TYPE, 8, 1>), dim3((mmax + 7) / 8, n), dim3(512), lds, st, g, x);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
int cdna4_launch_gemv_q_fused_grp(int type, const cdna4_gemv_group &g, int n, const float *x, hipStream_t st) {
    if (n < 1 || n > CDNA4_GEMV_GROUP_MAX) return cdna4_set_error_msg("gemv_q_fused_grp: 1 .. 4 matrices");
    if (!cdna4_gemv_fused_supported(type, g.K, 1) || ((uintptr_t)x & 15)) return cdna4_set_error_msg("gemv_q_fused_grp: unsupported shape or misaligned activation row");
    for (int i = 0; i < n; i++) {
        if (g.M[i] <= 0 || !g.W[i] || !g.Y[i]) return cdna4_set_error_msg("gemv_q_fused_grp: bad matrix");
        if (((uintptr_t)g.W[i] | (uintptr_t)g.w_row_bytes[i]) & ((type == CDNA4_Q4_K || type == CDNA4_Q5_K) ? 15 : 1)) return cdna4_set_error_msg("gemv_q_fused_grp: misaligned weight rows");
    }
    switch (type) {
        case CDNA4_Q4_K: return launch_fused_grp<CDNA4_Q4_K>(g, n, x, st);
        case CDNA4_Q5_K: return launch_fused_grp<CDNA4_Q5_K>(g, n, x, st);
        case CDNA4_Q6_K: return launch_fused_grp<CDNA4_Q6_K>(g, n, x, st);
        case CDNA4_Q4_0: return launch_fused_grp<CDNA4_Q4_0>(g, n, x, st);
        case CDNA4_Q8_0: return launch_fused_grp<CDNA4_Q8_0>(g, n, x, st);
    }
    return cdna4_set_error_msg("gemv_q_fused_grp: the five headline formats");
}
