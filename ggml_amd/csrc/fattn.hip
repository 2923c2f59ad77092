// fattn.hip — GGML_OP_FLASH_ATTN_EXT for F16 K / V, semantics of ggml_compute_forward_flash_attn_ext_f16
// (src/ggml-cpu/ggml-cpu.c:10805-11016): per query row, online softmax over the keys of
//     s = scale * <fp16(q), k>  [-> softcap * tanh(s)]  + slope(head) * mask[q][kv]
// and the softmax-weighted sum of the value rows.  Head sizes 64 / 128 / 256.
//
// One work-group = 32 query rows of one head; its four waves (one per SIMD) each take every fourth 32-key chunk and keep
// their own running (max, sum, O) — the partial results meet once at the end (in LDS, merged by wave 0).  Decode (1 query row) and
// prefill run the same code: with few query rows the four waves still split the keys.
//
// Everything is held in the accumulator layout of v_mfma_f32_32x32x16_f16 with the QUERY on the column (lane) axis:
//     S^T [32 kv x 32 q]  = K  [32 kv x hs]      . Q^T      A = 16-byte loads of K rows straight from HBM, B = fp16 Q held in registers
//     Vt  [32 kv x 32 d]  = V  [32 kv x 32 d]    . I        the matrix core as a transposer: A = 16-byte loads of V rows, B = 0/1 selection
//     O^T [32 d  x 32 q] += Vt^T[32 d x 32 kv]   . P^T      A = fp16(Vt) as it sits in the registers, B = fp16(P) as it sits in the registers
// Lane (q = lane % 32, h = lane / 32) holds accumulator rows rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15, of its column: the softmax
// statistics of a query row live in ONE lane pair (lane, lane ^ 32), rescaling O^T is a per-lane multiply, and the k-slots of the third
// product (8 h + j of step t <-> register 8 t + j) name the same key rho(8 t + j, h) on both operands because both came out of an
// accumulator whose rows are keys.  V is never gathered with 2-byte loads and nothing goes through LDS in the loop; the price is
// hs / 16 more MFMAs per chunk (V x identity is exact: products with 1.0, sums with 0.0, fp32 -> fp16 of an fp16 value).
// Precision: fp16 operands, fp32 accumulation (the CPU accumulates O in fp16 when V is F16: ours is the more accurate side).
#include "../../include/ggml_cdna4.h"
#include "cdna4_common.h"
#include "cdna4_kernels.h"
#include <math.h>

struct fattn_params {
    const char *q, *k, *v, *mask; float *dst;
    int64_t q_nb1, q_nb2, q_nb3, k_nb1, k_nb2, k_nb3, v_nb1, v_nb2, v_nb3, mask_nb1;       // bytes
    int n_q, n_head, n_kv, rk2, rk3, rv2, rv3;
    float scale, max_bias, logit_softcap, m0, m1; uint32_t n_head_log2;
};

template <int HS>
__global__ __launch_bounds__(256) void k_flash_attn_f16(const fattn_params p) {
    constexpr int NS = HS / 16, NB = HS / 32;                     // k-steps of Q.K, 32-wide blocks of the head dimension
    __shared__ float Os[HS * 32];                                 // one wave's O^T at a time: [d][q]
    __shared__ float Ms[32], Ss[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32, head = blockIdx.y, b3 = blockIdx.z;
    const int qi = min(q0 + n, p.n_q - 1);                        // rows past the end repeat the last one and are not stored

    // this lane's query row as fp16 B fragments: Q[qi][16 s + 8 h + e]  (q_to_vec_dot = fp32 -> fp16 row, ggml-cpu.c:10929)
    const float *qrow = (const float *)(p.q + (int64_t)qi * p.q_nb1 + (int64_t)head * p.q_nb2 + (int64_t)b3 * p.q_nb3);
    half8_t qf[NS];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int e = 0; e < 8; e++) qf[s][e] = (half_t)qrow[16 * s + 8 * h + e];
    // selection operands of the transposing product: B[k-slot 8 h + j][column n] = (n == 16 u + 8 h + j)
    half8_t sel[2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int j = 0; j < 8; j++) sel[u][j] = (n == 16 * u + 8 * h + j) ? (half_t)1.0f : (half_t)0.0f;

    const float slope = p.max_bias > 0.0f ? ((uint32_t)head < p.n_head_log2 ? powf(p.m0, (float)(head + 1)) : powf(p.m1, (float)(2 * (head - (int)p.n_head_log2) + 1))) : 1.0f;
    const char *kbase = p.k + (int64_t)(head / p.rk2) * p.k_nb2 + (int64_t)(b3 / p.rk3) * p.k_nb3;
    const char *vbase = p.v + (int64_t)(head / p.rv2) * p.v_nb2 + (int64_t)(b3 / p.rv3) * p.v_nb3;
    const half_t *mrow = p.mask ? (const half_t *)(p.mask + (int64_t)qi * p.mask_nb1) : nullptr;

    float M = -INFINITY, S = 0.0f;
    floatx16 o[NB];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[b][r] = 0.0f;

    const int nchunk = (p.n_kv + 31) / 32;
    for (int c = wave; c < nchunk; c += 4) {
        const int kv0 = 32 * c;
        const int64_t row = min(kv0 + n, p.n_kv - 1);             // A row n of this lane = key kv0 + n (past the end: repeated, masked below)
        // ---- S^T = K . Q^T
        const char *kp = kbase + row * p.k_nb1 + 16 * h;
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.0f;
#pragma unroll
        for (int st = 0; st < NS; st++) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8_t *>(kp + 32 * st), qf[st], s, 0, 0, 0);
        // ---- scale, softcap, mask; chunk maximum of this query row
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float x = s[r] * p.scale;
            if (p.logit_softcap != 0.0f) x = p.logit_softcap * tanhf(x);
            if (kv < p.n_kv) { if (mrow) x += slope * (float)mrow[kv]; } else x = -INFINITY;
            s[r] = x; mx = fmaxf(mx, x);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float Mn = fmaxf(M, mx);
        // everything masked so far: keep (M, S, O) = (-inf, 0, 0) — the CPU skips -inf entries (ggml-cpu.c:10935-10938)
        const float ms = (M == -INFINITY) ? 0.0f : expf(M - Mn);
        float sum = 0.0f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float e = (Mn == -INFINITY) ? 0.0f : expf(s[r] - Mn);
            sum += e; pf[r >> 3][r & 7] = (half_t)e;
        }
        sum += __shfl_xor(sum, 32);
        S = S * ms + sum; M = Mn;
        // ---- O^T = O^T * ms + Vt^T . P^T, one 32-wide block of the head dimension at a time
        const char *vp = vbase + row * p.v_nb1 + 16 * h;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            floatx16 vt;
#pragma unroll
            for (int r = 0; r < 16; r++) vt[r] = 0.0f;
            vt = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8_t *>(vp + 64 * b), sel[0], vt, 0, 0, 0);
            vt = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8_t *>(vp + 64 * b + 32), sel[1], vt, 0, 0, 0);
            half8_t vf[2];
#pragma unroll
            for (int r = 0; r < 16; r++) { vf[r >> 3][r & 7] = (half_t)vt[r]; o[b][r] *= ms; }
            o[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], pf[0], o[b], 0, 0, 0);
            o[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1], pf[1], o[b], 0, 0, 0);
        }
    }

    // ---- the four partial results meet: waves 1..3 hand theirs to wave 0 one after the other through LDS (same lane <-> element
    // mapping on both sides: Os[d][q], q = lane % 32 — conflict-free)
    for (int w = 1; w < 4; w++) {
        __syncthreads();
        if (wave == w) {
            if (h == 0) { Ms[n] = M; Ss[n] = S; }
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) Os[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = o[b][r];
        }
        __syncthreads();
        if (wave == 0) {
            const float Mw = Ms[n], Sw = Ss[n], Mn = fmaxf(M, Mw);
            const float a0 = (M == -INFINITY) ? 0.0f : expf(M - Mn), aw = (Mw == -INFINITY) ? 0.0f : expf(Mw - Mn);
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[b][r] = o[b][r] * a0 + Os[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] * aw;
            S = S * a0 + Sw * aw; M = Mn;
        }
    }
    // ---- V /= S; dst is [hs, n_head, n_q, batch] (the permute(0, 2, 1, 3) of ggml-cpu.c:11012)
    if (wave == 0 && q0 + n < p.n_q) {
        const float inv = 1.0f / S;
        float *out = p.dst + (((int64_t)b3 * p.n_q + (q0 + n)) * p.n_head + head) * HS;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(out + 32 * b + 8 * g + 4 * h) = make_float4(o[b][4 * g] * inv, o[b][4 * g + 1] * inv, o[b][4 * g + 2] * inv, o[b][4 * g + 3] * inv);
    }
}

#define NEED(cond, msg) do { if (!(cond)) return cdna4_set_error_msg(msg); } while (0)
typedef ggml_cdna4_tensor T4;

extern "C" int ggml_cdna4_op_flash_attn_ext_supported(int64_t head_size, int kv_type) {
    return (head_size == 64 || head_size == 128 || head_size == 256) && kv_type == CDNA4_F16;
}

extern "C" int ggml_cdna4_op_flash_attn_ext(const T4 *q, const T4 *k, const T4 *v, const T4 *mask, const T4 *d,
                                            float scale, float max_bias, float logit_softcap, void *stream) {
    NEED(q && k && v && d, "flash_attn_ext: q, k, v and dst are required");
    NEED(q->type == CDNA4_F32 && d->type == CDNA4_F32 && k->type == CDNA4_F16 && v->type == CDNA4_F16, "flash_attn_ext: F32 q / dst and F16 k / v only");
    const int64_t D = q->ne[0], N = q->ne[1], H = q->ne[2], B3 = q->ne[3], KV = k->ne[1];
    NEED(ggml_cdna4_op_flash_attn_ext_supported(D, k->type), "flash_attn_ext: head size must be 64, 128 or 256");
    NEED(k->ne[0] == D && v->ne[0] == D && v->ne[1] == KV && k->ne[2] == v->ne[2] && k->ne[3] == v->ne[3], "flash_attn_ext: k / v shape mismatch");
    NEED(k->ne[2] > 0 && k->ne[3] > 0 && H % k->ne[2] == 0 && B3 % k->ne[3] == 0, "flash_attn_ext: heads / batch not broadcastable over k / v");
    NEED(d->ne[0] == D && d->ne[1] == H && d->ne[2] == N && d->ne[3] == B3, "flash_attn_ext: dst must be [head_size, n_head, n_q, batch]");
    NEED(d->nb[0] == 4 && d->nb[1] == 4 * D && d->nb[2] == d->nb[1] * H && d->nb[3] == d->nb[2] * N, "flash_attn_ext: dst must be contiguous");
    NEED(q->nb[0] == 4 && k->nb[0] == 2 && v->nb[0] == 2, "flash_attn_ext: rows must be contiguous");
    NEED(!(((uintptr_t)k->data | (uintptr_t)v->data | (uintptr_t)d->data | (uintptr_t)k->nb[1] | (uintptr_t)k->nb[2] | (uintptr_t)k->nb[3] |
            (uintptr_t)v->nb[1] | (uintptr_t)v->nb[2] | (uintptr_t)v->nb[3]) & 15), "flash_attn_ext: k / v rows and dst must be 16-byte aligned");
    NEED(!(((uintptr_t)q->data | (uintptr_t)q->nb[1] | (uintptr_t)q->nb[2] | (uintptr_t)q->nb[3]) & 3), "flash_attn_ext: q must be 4-byte aligned");
    if (mask) {
        NEED(mask->type == CDNA4_F16 && mask->nb[0] == 2 && mask->ne[0] == KV && mask->ne[1] >= N && mask->ne[2] == 1 && mask->ne[3] == 1, "flash_attn_ext: mask must be F16 [n_kv, >= n_q]");
    } else NEED(max_bias == 0.0f, "flash_attn_ext: ALiBi needs a mask");
    if (N <= 0 || H <= 0 || B3 <= 0 || D <= 0) return 0;
    NEED(KV > 0, "flash_attn_ext: no keys");
    NEED(H <= 65535 && B3 <= 65535, "flash_attn_ext: too many heads / batches for one grid");

    fattn_params p{};
    p.q = (const char *)q->data; p.k = (const char *)k->data; p.v = (const char *)v->data; p.mask = mask ? (const char *)mask->data : nullptr; p.dst = (float *)d->data;
    p.q_nb1 = q->nb[1]; p.q_nb2 = q->nb[2]; p.q_nb3 = q->nb[3];
    p.k_nb1 = k->nb[1]; p.k_nb2 = k->nb[2]; p.k_nb3 = k->nb[3]; p.v_nb1 = v->nb[1]; p.v_nb2 = v->nb[2]; p.v_nb3 = v->nb[3];
    p.mask_nb1 = mask ? mask->nb[1] : 0;
    p.n_q = (int)N; p.n_head = (int)H; p.n_kv = (int)KV;
    p.rk2 = (int)(H / k->ne[2]); p.rk3 = (int)(B3 / k->ne[3]); p.rv2 = (int)(H / v->ne[2]); p.rv3 = (int)(B3 / v->ne[3]);
    // ggml-cpu.c:10876-10884
    p.scale = logit_softcap != 0.0f ? scale / logit_softcap : scale; p.max_bias = max_bias; p.logit_softcap = logit_softcap;
    p.n_head_log2 = 1u << (uint32_t)floorf(log2f((float)H));
    p.m0 = powf(2.0f, -(max_bias) / p.n_head_log2); p.m1 = powf(2.0f, -(max_bias / 2.0f) / p.n_head_log2);

    const dim3 grid((unsigned)((N + 31) / 32), (unsigned)H, (unsigned)B3);
    hipStream_t st = (hipStream_t)stream;
    if (D == 64) hipLaunchKernelGGL(k_flash_attn_f16<64>, grid, dim3(256), 0, st, p);
    else if (D == 128) hipLaunchKernelGGL(k_flash_attn_f16<128>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_flash_attn_f16<256>, grid, dim3(256), 0, st, p);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
