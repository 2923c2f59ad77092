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
#include <stdlib.h>

__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

template <int TYPE, int NB> struct Unit;

// ---- Q4_K: unit = (superblock, 64-group g): 32 bytes of nibbles, low -> k 64g+l, high -> 64g+32+l ---------
template <int NB> struct Unit<CDNA4_Q4_K, NB> {
    static constexpr int UK = 64;
    __device__ static void dot(const uint8_t *wrow, int u, const cdna4_gemv_args &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, g = u & 3;
        const uint8_t *blk = wrow + (int64_t)sb * 144;
        const u32x4 hdr = ld_u32x4(blk);
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        int sc_lo, m_lo, sc_hi, m_hi;
        k4_scale_min_rt(hdr.y, hdr.z, hdr.w, 2 * g, sc_lo, m_lo);
        k4_scale_min_rt(hdr.y, hdr.z, hdr.w, 2 * g + 1, sc_hi, m_hi);
        const u32x4 q0 = ld_u32x4(blk + 16 + 32 * g), q1 = ld_u32x4(blk + 32 + 32 * g);
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *y = a.qs + (int64_t)col[c] * a.K + sb * 256 + 64 * g;
            const u32x4 y0 = ld_u32x4(y), y1 = ld_u32x4(y + 16), y2 = ld_u32x4(y + 32), y3 = ld_u32x4(y + 48);
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
};

// ---- Q5_K: Q4_K plus a fifth bit per weight from qh[l] bit 2g (low) / 2g+1 (high) ---------------------------
template <int NB> struct Unit<CDNA4_Q5_K, NB> {
    static constexpr int UK = 64;
    __device__ static void dot(const uint8_t *wrow, int u, const cdna4_gemv_args &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, g = u & 3;
        const uint8_t *blk = wrow + (int64_t)sb * 176;
        const u32x4 hdr = ld_u32x4(blk);
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        int sc_lo, m_lo, sc_hi, m_hi;
        k4_scale_min_rt(hdr.y, hdr.z, hdr.w, 2 * g, sc_lo, m_lo);
        k4_scale_min_rt(hdr.y, hdr.z, hdr.w, 2 * g + 1, sc_hi, m_hi);
        const u32x4 h0 = ld_u32x4(blk + 16), h1 = ld_u32x4(blk + 32);
        const u32x4 q0 = ld_u32x4(blk + 48 + 32 * g), q1 = ld_u32x4(blk + 64 + 32 * g);
        const uint32_t qh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        uint32_t wl[8], wh[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            wl[i] = (w[i] & 0x0F0F0F0Fu) | (((qh[i] >> (2 * g)) & 0x01010101u) << 4);
            wh[i] = ((w[i] >> 4) & 0x0F0F0F0Fu) | (((qh[i] >> (2 * g + 1)) & 0x01010101u) << 4);
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *y = a.qs + (int64_t)col[c] * a.K + sb * 256 + 64 * g;
            const u32x4 y0 = ld_u32x4(y), y1 = ld_u32x4(y + 16), y2 = ld_u32x4(y + 32), y3 = ld_u32x4(y + 48);
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
};

// ---- Q6_K: unit = (superblock, half n, 16-lane slice lb): 16 values of l -> 4 x 16 weights at
//      k = 128n + 32*quad + 16lb + i, int8 scale per 16 (src/ggml-quants.c:1690-1719) -----------------------
template <int NB> struct Unit<CDNA4_Q6_K, NB> {
    static constexpr int UK = 64;
    __device__ static void dot(const uint8_t *wrow, int u, const cdna4_gemv_args &a, const int (&col)[NB], float (&acc)[NB]) {
        const int sb = u >> 2, n = (u >> 1) & 1, lb = u & 1;
        const uint8_t *blk = wrow + (int64_t)sb * 210;                       // 2-byte aligned only
        const uint8_t *ql_a = blk + 64 * n + 16 * lb, *ql_b = ql_a + 32, *qhp = blk + 128 + 32 * n + 16 * lb;
        const int8_t *scp = reinterpret_cast<const int8_t *>(blk + 192 + 8 * n + lb);
        const float d = h2f(ld_u16(blk + 208));
        uint32_t q[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t la = ld_u32_a2(ql_a + 4 * i), lv = ld_u32_a2(ql_b + 4 * i), hq = ld_u32_a2(qhp + 4 * i);
            q[0][i] = (la & 0x0F0F0F0Fu) | ((hq & 0x03030303u) << 4);
            q[1][i] = (lv & 0x0F0F0F0Fu) | (((hq >> 2) & 0x03030303u) << 4);
            q[2][i] = ((la >> 4) & 0x0F0F0F0Fu) | (((hq >> 4) & 0x03030303u) << 4);
            q[3][i] = ((lv >> 4) & 0x0F0F0F0Fu) | (((hq >> 6) & 0x03030303u) << 4);
        }
        const int sc[4] = {scp[0], scp[2], scp[4], scp[6]};
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *y = a.qs + (int64_t)col[c] * a.K + sb * 256 + 128 * n + 16 * lb;
            const int16_t *bs = a.bsums + (int64_t)col[c] * (a.K / 16) + sb * 16 + 8 * n + lb;
            int isum = 0;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const u32x4 yv = ld_u32x4(y + 32 * qd);
                int s = dot4(q[qd][0], yv.x, 0); s = dot4(q[qd][1], yv.y, s); s = dot4(q[qd][2], yv.z, s); s = dot4(q[qd][3], yv.w, s);
                isum += sc[qd] * (s - 32 * (int)bs[2 * qd]);                // sum (q-32)*y = sum q*y - 32*bsum
            }
            const float yd = a.d[(int64_t)col[c] * (a.K / 256) + sb];
            acc[c] += (d * yd) * (float)isum;
        }
    }
};

// ---- Q4_0: unit = 18-byte block, nibble j -> k j (low), j+16 (high), value q-8 ------------------------------
template <int NB> struct Unit<CDNA4_Q4_0, NB> {
    static constexpr int UK = 32;
    __device__ static void dot(const uint8_t *wrow, int u, const cdna4_gemv_args &a, const int (&col)[NB], float (&acc)[NB]) {
        const uint8_t *blk = wrow + (int64_t)u * 18;
        const float d = h2f(ld_u16(blk));
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = ld_u32_a2(blk + 2 + 4 * i);
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *y = a.qs + (int64_t)col[c] * a.K + u * 32;
            const u32x4 y0 = ld_u32x4(y), y1 = ld_u32x4(y + 16);
            const uint32_t yl[4] = {y0.x, y0.y, y0.z, y0.w}, yh[4] = {y1.x, y1.y, y1.z, y1.w};
            int s = 0, ys = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s = dot4(w[i] & 0x0F0F0F0Fu, yl[i], s); s = dot4((w[i] >> 4) & 0x0F0F0F0Fu, yh[i], s);
                ys = dot4(0x01010101u, yl[i], ys); ys = dot4(0x01010101u, yh[i], ys);
            }
            acc[c] += (float)(s - 8 * ys) * d * a.d[(int64_t)col[c] * (a.K / 32) + u];
        }
    }
};

// ---- Q8_0: unit = 34-byte block ----------------------------------------------------------------------------
template <int NB> struct Unit<CDNA4_Q8_0, NB> {
    static constexpr int UK = 32;
    __device__ static void dot(const uint8_t *wrow, int u, const cdna4_gemv_args &a, const int (&col)[NB], float (&acc)[NB]) {
        const uint8_t *blk = wrow + (int64_t)u * 34;
        const float d = h2f(ld_u16(blk));
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = ld_u32_a2(blk + 2 + 4 * i);
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int8_t *y = a.qs + (int64_t)col[c] * a.K + u * 32;
            const u32x4 y0 = ld_u32x4(y), y1 = ld_u32x4(y + 16);
            const uint32_t yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            int s = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) s = dot4(w[i], yv[i], s);
            acc[c] += (float)s * (d * a.d[(int64_t)col[c] * (a.K / 32) + u]);
        }
    }
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
        for (int c = 0; c < NB; c++) { col[c] = blockIdx.y * NB + c; ycol[c] = col[c]; }
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
            const float s = wave_sum(acc[r][c]);
            if (lane == 0 && row0 + r < a.M) a.Y[(int64_t)ycol[c] * a.y_col_stride + row0 + r] = s;
        }
}

template <int TYPE, int NB>
static void launch_nb(const cdna4_gemv_args &a, hipStream_t st) {
    // one row per wave.  (4 rows per wave — fewer, fatter waves sharing the activation loads — measured SLOWER on
    // MI355X: 6.7 vs 5.3 us cold at 4096x4096; CDNA4_GEMV_ROWS=2 selects the 2-row variant for experiments.)
    static const int rows_env = getenv("CDNA4_GEMV_ROWS") ? atoi(getenv("CDNA4_GEMV_ROWS")) : 1;
    if (rows_env == 2 && NB <= 2) hipLaunchKernelGGL((k_gemv_q<TYPE, NB, false, 2>), dim3((a.M + 7) / 8, 1), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemv_q<TYPE, NB, false, 1>), dim3((a.M + 3) / 4, 1), dim3(256), 0, st, a);
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
        switch (nb) {
            case 1: launch_nb<TYPE, 1>(a, st); break; case 2: launch_nb<TYPE, 2>(a, st); break;
            case 3: launch_nb<TYPE, 3>(a, st); break; case 4: launch_nb<TYPE, 4>(a, st); break;
            case 5: launch_nb<TYPE, 5>(a, st); break; case 6: launch_nb<TYPE, 6>(a, st); break;
            case 7: launch_nb<TYPE, 7>(a, st); break; default: launch_nb<TYPE, 8>(a, st); break;
        }
        CDNA4_CHECK_LAUNCH();
    }
    return 0;
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
    }
    return cdna4_set_error_msg("gemv_q: unsupported weight type");
}
