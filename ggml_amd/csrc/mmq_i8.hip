// mmq_i8.hip — small batches (2 .. 64 activation rows) of the quantized MUL_MAT on the INT8 matrix cores.
//
// Between the one-row decode GEMV and the fp16 prefill GEMM there was nothing matrix-shaped: 2..8 rows ran the v_dot4_i32_i8 GEMV with eight
// columns per weight unit (27 us at 4096 x 14336 x 8 where one pass over the weights takes 10), 9..64 rows a mostly empty 128-row fp16 tile
// (31-35 us).  This kernel computes exactly what ggml_vec_dot_q4_K_q8_K computes (src/ggml-cpu/ggml-cpu-quants.c:5549-6194) — the INTEGER block
// dot products of the 4-bit weights with the Q8_K-quantized activations, then the fp32 scale products per superblock — but sixteen activation rows
// by sixteen weight rows by one 32-weight sub-block per instruction: v_mfma_i32_16x16x32_i8.  The integer sums are exact, so the result agrees
// with the GEMV units (and the CPU) to fp32 summation order, not to the fp16 rounding of the prefill GEMM.
//   reference dispatcher being replaced: ggml-cuda picks mul_mat_vec_q up to 8 columns and mmq above (src/ggml-cuda/ggml-cuda.cu:1844-1905, mmvq.cuh:3)
//
// MI355X mapping.  HBM-bound like the GEMV (each weight byte is read once); the matrix core only has to keep up.
//   * work-group = 16 weight rows, 8 waves; wave w takes the superblocks sb = w, w + 8, ... of those rows (the K split stays inside the
//     work-group: partial 16 x NB tiles meet in LDS at the end) — 256 work-groups at M = 4096, one per CU, like the one-launch decode kernel;
//   * MFMA operands: A = activations (rows = b), B = weights (columns = m).  Lane l supplies, for BOTH, row / column l % 16 and the eight
//     consecutive k of group l / 16 — so a lane holds the bytes of ITS weight row and decodes that row's 6-bit scales itself, and the
//     accumulator lane (column m = l % 16, rows b = 4 (l / 16) + i) needs nothing from other lanes on the weight side;
//   * weights come straight from global memory in that layout: per 64-weight group one 8-byte load per lane (low nibbles = sub-block 2g,
//     high = 2g + 1: two MFMAs per load), the 16-byte header once per superblock (the four lanes of a row load the same bytes: one L1 line);
//     the next superblock's loads are issued before the current one's arithmetic;
//   * activation side per superblock: eight 8-byte fragments per 16-column group, and per wave the sixteen columns' d and bsums pair sums staged
//     through a private LDS slot (lane (b, part) loads four bsums, adds them in pairs) for the minimum term sum_j m_j (bsums[2j] + bsums[2j+1]);
//   * per superblock and lane: sumi[i] = sum_j sc_j S_j[i] and summs[i] = sum_j m_j bs_j[i] in int32 (v_mad_i32_i24), then
//     acc[i] += (d dy_i) sumi[i] - (dmin dy_i) summs[i] in fp32 — the association of ggml_vec_dot_q4_K_q8_K's scalar body.
#include "cdna4_common.h"
#include "cdna4_kernels.h"
#include "epilogue.h"

typedef int intx4 __attribute__((ext_vector_type(4)));

struct mmq_args {
    const uint8_t *W; int64_t w_row_bytes;
    const int8_t *qs; const float *d; const int16_t *bsums;           // Q8_K activations (quantize_act.hip): [B][K], [B][K/256], [B][K/16]
    float *Y; int64_t y_row;                                          // Y[b * y_row + m]
    int M, K, B;
    cdna4_epilogue epi;
};

__device__ __forceinline__ u32x2 ld_u32x2_a4(const void *p) { return *reinterpret_cast<const u32x2 *>(p); }
// 24-bit integer multiply (v_mul_i32_i24: full rate; the 32-bit v_mul_lo_u32 is a quarter of it) — both factors fit by construction here
#if defined(__HIPCC__)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
#else
static inline int mul24(int a, int b) { return a * b; }
#endif
__device__ __forceinline__ long as_i64(uint32_t lo, uint32_t hi) { return (long)(((uint64_t)hi << 32) | lo); }

// NCG column groups of 16 activation rows each (B <= 16 NCG)
template <int NCG>
__global__ __launch_bounds__(512) void k_mmq_q4_K(const mmq_args a) {
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) uint8_t smem[NW * NCG * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, grp = lane >> 4;                        // MFMA lane roles: row / column l % 16, k-group l / 16
    const int m0 = blockIdx.x * 16;
    const int mrow = min(m0 + col, a.M - 1);
    const uint8_t *wrow = a.W + (int64_t)mrow * a.w_row_bytes;
    const int nsb = a.K / 256;
    // per wave and column group: [16 columns][8 pair sums int16 | float d | pad] = 32 bytes per column (16-byte aligned reads)
    uint8_t *meta = smem + wave * (NCG * 16 * 32);

    float acc[NCG][4];
#pragma unroll
    for (int g = 0; g < NCG; g++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[g][i] = 0.f;

    // everything a superblock needs from memory, requested one superblock AHEAD (cur / nxt): this row's weights, and per 16-column group the eight
    // activation fragments, four bsums and the column's d — the first version loaded the activation side at its use and spent a full L2 round trip
    // per fragment (24 us at 4096 x 14336 x 16 where the weights stream in 5)
    struct WSb { u32x4 hdr; u32x2 q[4]; };
    struct XSb { u32x2 xl[4], xh[4], bs; float dy; };
    auto load_w = [&](int sb) __attribute__((always_inline)) {
        WSb w; const uint8_t *blk = wrow + (int64_t)sb * 144;
        w.hdr = ld_u32x4(blk);
#pragma unroll
        for (int g = 0; g < 4; g++) w.q[g] = ld_u32x2_a4(blk + 16 + 32 * g + 8 * grp);
        return w;
    };
    auto load_x = [&](int sb, int g) __attribute__((always_inline)) {
        XSb x; const int b = min(g * 16 + col, a.B - 1);
        const int8_t *xq = a.qs + (int64_t)b * a.K + sb * 256 + 8 * grp;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) { x.xl[gq] = ld_u32x2_a4(xq + 64 * gq); x.xh[gq] = ld_u32x2_a4(xq + 64 * gq + 32); }
        x.bs = ld_u32x2_a4(a.bsums + (int64_t)b * (a.K / 16) + sb * 16 + 4 * grp);
        x.dy = a.d[(int64_t)b * nsb + sb];
        return x;
    };
    WSb cur{}; XSb xc[NCG];
#pragma unroll
    for (int g = 0; g < NCG; g++) xc[g] = XSb{};
    if (wave < nsb) {
        cur = load_w(wave);
#pragma unroll
        for (int g = 0; g < NCG; g++) xc[g] = load_x(wave, g);
    }
    for (int sb = wave; sb < nsb; sb += NW) {
        WSb nxt{}; XSb xn[NCG];
#pragma unroll
        for (int g = 0; g < NCG; g++) xn[g] = XSb{};
        if (sb + NW < nsb) {
            nxt = load_w(sb + NW);
#pragma unroll
            for (int g = 0; g < NCG; g++) xn[g] = load_x(sb + NW, g);
        }
        // ---- activation metadata of this superblock -> the wave's LDS slot: lane (b = col, part = grp) adds bsums 4 part .. 4 part + 3 in pairs
        CDNA4_WAVE_LDS_SYNC();                                           // the previous superblock's reads of the slot are over
#pragma unroll
        for (int g = 0; g < NCG; g++) {
            const u32x2 bs = xc[g].bs;
            const int p0 = (int)(int16_t)(bs.x & 0xFFFF) + (int)(int16_t)(bs.x >> 16), p1 = (int)(int16_t)(bs.y & 0xFFFF) + (int)(int16_t)(bs.y >> 16);
            *reinterpret_cast<uint32_t *>(meta + (g * 16 + col) * 32 + 4 * grp) = (uint32_t)(p0 & 0xFFFF) | ((uint32_t)p1 << 16);
            if (grp == 0) *reinterpret_cast<float *>(meta + (g * 16 + col) * 32 + 16) = xc[g].dy;
        }
        CDNA4_WAVE_LDS_SYNC();                                           // the slot is written: other lanes' entries may be read
        // ---- this row's scales and minima (get_scale_min_k4, src/ggml-quants.c:631-638), d and dmin
        const float dw = h2f(cur.hdr.x & 0xFFFF), dmin = h2f(cur.hdr.x >> 16);
        int sc[8], mn[8];
#pragma unroll
        for (int j = 0; j < 8; j++) k4_scale_min_rt(cur.hdr.y, cur.hdr.z, cur.hdr.w, j, sc[j], mn[j]);
#pragma unroll
        for (int g = 0; g < NCG; g++) {
            intx4 sumi = {0, 0, 0, 0};
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {                             // 64-weight group gq: sub-blocks 2 gq (low nibbles) and 2 gq + 1 (high)
                const u32x2 xl = xc[g].xl[gq], xh = xc[g].xh[gq];
                const intx4 z = {0, 0, 0, 0};
                const intx4 sl = __builtin_amdgcn_mfma_i32_16x16x32_i8(as_i64(xl.x, xl.y), as_i64(cur.q[gq].x & 0x0F0F0F0Fu, cur.q[gq].y & 0x0F0F0F0Fu), z, 0, 0, 0);
                const intx4 sh = __builtin_amdgcn_mfma_i32_16x16x32_i8(as_i64(xh.x, xh.y), as_i64((cur.q[gq].x >> 4) & 0x0F0F0F0Fu, (cur.q[gq].y >> 4) & 0x0F0F0F0Fu), z, 0, 0, 0);
                // |S| <= 32 * 15 * 127 and sc < 64: 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter of it)
#pragma unroll
                for (int i = 0; i < 4; i++) sumi[i] += mul24(sc[2 * gq], sl[i]) + mul24(sc[2 * gq + 1], sh[i]);
            }
            // minimum term and the fp32 scale products: rows b = 4 grp + i of this column group
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint8_t *mb = meta + (g * 16 + 4 * grp + i) * 32;
                const u32x4 ps = *reinterpret_cast<const u32x4 *>(mb);
                const float dy = *reinterpret_cast<const float *>(mb + 16);
                const uint32_t pw[4] = {ps.x, ps.y, ps.z, ps.w};
                int summs = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) summs += mul24(mn[2 * j], (int)(int16_t)(pw[j] & 0xFFFF)) + mul24(mn[2 * j + 1], (int)(int16_t)(pw[j] >> 16));
                acc[g][i] += (dw * dy) * (float)sumi[i] - (dmin * dy) * (float)summs;
            }
        }
        cur = nxt;
#pragma unroll
        for (int g = 0; g < NCG; g++) xc[g] = xn[g];
    }
    // ---- the eight waves' partial tiles meet in LDS: [wave][group][4][64 lanes]
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int g = 0; g < NCG; g++)
#pragma unroll
        for (int i = 0; i < 4; i++) red[((wave * NCG + g) * 4 + i) * 64 + lane] = acc[g][i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int g = 0; g < NCG; g++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w++) s += red[((w * NCG + g) * 4 + i) * 64 + lane];
                const int b = g * 16 + 4 * grp + i, m = m0 + col;
                if (b < a.B && m < a.M) a.Y[(int64_t)b * a.y_row + m] = epilogue_apply(a.epi, s, m, b);
            }
    }
}

bool cdna4_mmq_supported(int type, int64_t M, int64_t K, int64_t B) {
    return type == CDNA4_Q4_K && M > 0 && K >= 256 && K % 256 == 0 && B >= 2 && B <= 64;
}
// a.qs / a.d / a.bsums: the Q8_K workspace ggml_cdna4_prepare_act fills (path GEMV)
int cdna4_launch_mmq(const cdna4_gemv_args &g, hipStream_t st) {
    if (!cdna4_mmq_supported(g.type, g.M, g.K, g.ncol) || g.ids) return cdna4_set_error_msg("mmq: unsupported type / shape");
    if (((uintptr_t)g.W | (uintptr_t)g.w_row_bytes) & 15) return cdna4_set_error_msg("mmq: Q4_K rows must be 16-byte aligned");
    if (((uintptr_t)g.qs | (uintptr_t)g.bsums) & 7) return cdna4_set_error_msg("mmq: quantized activations must be 8-byte aligned");
    mmq_args a{};
    a.W = g.W; a.w_row_bytes = g.w_row_bytes; a.qs = g.qs; a.d = g.d; a.bsums = g.bsums; a.Y = g.Y; a.y_row = g.y_col_stride;
    a.M = g.M; a.K = g.K; a.B = g.ncol; a.epi = g.epi;
    const dim3 grid((g.M + 15) / 16);
    const int ncg = (g.ncol + 15) / 16;
    if (ncg == 1) hipLaunchKernelGGL(k_mmq_q4_K<1>, grid, dim3(512), 0, st, a);
    else if (ncg == 2) hipLaunchKernelGGL(k_mmq_q4_K<2>, grid, dim3(512), 0, st, a);
    else if (ncg == 3) hipLaunchKernelGGL(k_mmq_q4_K<3>, grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(k_mmq_q4_K<4>, grid, dim3(512), 0, st, a);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
