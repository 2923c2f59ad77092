// cdna4_kernels.h — internal launcher prototypes (C++), one per .hip file.  The public C-ABI is
// include/ggml_cdna4.h, implemented in capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// quantize_act.hip
int cdna4_launch_quantize_q8_K(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d,
                               int16_t *bsums, void *xh, hipStream_t st);
int cdna4_launch_quantize_q8_0(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d,
                               void *xh, bool ref_rounding, hipStream_t st);

// grouped MUL_MAT_ID (quantize_act.hip): device-side counting sort of the expert ids + gathering activation quantizer
int cdna4_launch_moe_plan(const int32_t *ids, int64_t ids_tok_stride, int n_tok, int n_used, int n_b, int n_expert, int img_rows,
                          int32_t *img_src, int32_t *img_dst, int32_t *tile_expert, hipStream_t st);
// grouped MUL_MAT_ID, stream-k form (round 6): ONE launch = the planner (work-group 0) + the activation quantizer in token order; then k_gemm_kq_sk (gemm_q_sk.hip)
#define CDNA4_SK_REC 260               // int32 per tile record: expert, rows, index within the expert, fragments, src row [128], dst pair [128]
#define CDNA4_SK_MAX_TILES 2047        // tiles the planner's LDS tables hold
int cdna4_launch_moe_sk_front(const int32_t *ids, int64_t ids_tok_stride, int n_tok, int n_used, int n_b, int n_expert, int ntile_cap, int upt, int G, const int *cw,
                              int32_t *tile_rec, int32_t *wg_begin, const float *x, int64_t x_row_stride, int64_t K, bool kq, void *xh, hipStream_t st);
// Q8_1 (activations of Q4_1 / Q5_1 weights): as q8_0 without ref_rounding, plus s[B][K/32] = fp16(d * sum of the block's quants) as fp32
// two_part: xh has 2 K columns, the second K holding s in the first column of every 32-block and zeros elsewhere (the GEMM image of Q4_1 / Q5_1)
int cdna4_launch_quantize_q8_1(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d, float *s, void *xh, bool two_part, hipStream_t st);
int cdna4_launch_quantize_q8_K_gather(const float *x, int64_t x_row_stride, int64_t K, int64_t img_rows, const int32_t *src_rows, void *xh, hipStream_t st);

// gemv_q.hip — int8-dot decode path.  Activations are the SoA workspace of quantize_act.hip.
// the element-wise tail a MUL_MAT may carry (epilogue.h): + bias[m], GELU, + resid[b][m] — in that order, each a separate fp32 operation
struct cdna4_epilogue {
    const float *bias;                 // [M] or nullptr
    const float *resid;                // [B][resid_row_stride], element (b, m) at b * resid_row_stride + m, or nullptr
    int64_t resid_row_stride;
    int act;                           // 0 none, 1 GELU (after the bias, before the residual)
};

struct cdna4_gemv_args {
    int type;
    const uint8_t *W; int64_t w_row_bytes;          // M rows of K weights
    const int8_t *qs; const float *d; const int16_t *bsums;   // [ncol][K], [ncol][K/QKA], [ncol][K/16]
    float *Y; int64_t y_col_stride;                 // Y[c * y_col_stride + m]
    int M, K, ncol;
    // MUL_MAT_ID: if ids != nullptr, column c = (token t, slot u): expert = ids[t*ids_tok_stride + u],
    // W += expert * w_expert_bytes, activation column = t * n_b + (u % n_b)
    const int32_t *ids; int64_t ids_tok_stride; int64_t w_expert_bytes; int n_used, n_b, n_expert;
    cdna4_epilogue epi;                             // applied to every stored value (zeroed = none); not with ids
};
// Y[b][m] = epilogue(Y[b][m]) in place — the tail of a GEMM-path MUL_MAT, one launch instead of two or three (ops.hip)
int cdna4_launch_epilogue(float *Y, int64_t y_row_stride, int64_t M, int64_t B, const cdna4_epilogue &e, hipStream_t st);
// to_float of a block-quantized tensor (rows contiguous, any row / batch strides) rounded to fp16, written in logical order as
// [ne3][ne2][ne1] rows dst_row >= ne0 halves apart — the K / V operands of FLASH_ATTN_EXT when they arrive quantized (ops.hip)
struct ggml_cdna4_tensor;
bool cdna4_to_f16_dense_supported(int type);
int cdna4_launch_to_f16_dense(const ggml_cdna4_tensor *a, void *dst, int64_t dst_row, hipStream_t st);
int cdna4_launch_gemv_q(const cdna4_gemv_args &a, hipStream_t st);
// single-column decode with the activation quantizer fused in (x = fp32 row; a.qs/d/bsums unused)
bool cdna4_gemv_fused_supported(int type, int64_t K, int64_t B);
int cdna4_launch_gemv_q_fused(const cdna4_gemv_args &a, const float *x, hipStream_t st);
// 2..8 activation rows in one launch (a.ncol rows of x, x_row_stride elements apart; the five main formats)
int cdna4_launch_gemv_q_fused_n(const cdna4_gemv_args &a, const float *x, int64_t x_row_stride, hipStream_t st);
// 2..8 pre-quantized activation rows (a.qs / a.d / a.bsums) copied into LDS once per work-group
bool cdna4_gemv_staged_supported(int type, int64_t K, int64_t B);
int cdna4_launch_gemv_q_staged(const cdna4_gemv_args &a, hipStream_t st);
// several matrices (one type, one K) against ONE activation row in one launch: bit-identical to the separate one-row calls
#define CDNA4_GEMV_GROUP_MAX 4
struct cdna4_gemv_group { const uint8_t *W[CDNA4_GEMV_GROUP_MAX]; int64_t w_row_bytes[CDNA4_GEMV_GROUP_MAX]; float *Y[CDNA4_GEMV_GROUP_MAX]; const float *bias[CDNA4_GEMV_GROUP_MAX]; int M[CDNA4_GEMV_GROUP_MAX]; int K; };
int cdna4_launch_gemv_q_fused_grp(int type, const cdna4_gemv_group &g, int n, const float *x, hipStream_t st);
// single-token MUL_MAT_ID in one launch (a.ids set, a.ncol = n_used slots; x rows x_row_stride apart, slot u reads row u % a.n_b)
int cdna4_launch_gemv_q_fused_ids(const cdna4_gemv_args &a, const float *x, int64_t x_row_stride, hipStream_t st);

// mmq_i8.hip — 2 .. 64 activation rows on the int8 matrix cores (v_mfma_i32_16x16x32_i8): the GEMV's integer block dots, sixteen columns at a time;
// a.qs / a.d / a.bsums as for cdna4_launch_gemv_q (Q8_K workspace), a.epi applied in the store
bool cdna4_mmq_supported(int type, int64_t M, int64_t K, int64_t B);
int cdna4_launch_mmq(const cdna4_gemv_args &a, hipStream_t st);
bool cdna4_mmq_ids_supported(int type, int64_t K);                  // grouped MUL_MAT_ID on the int8 matrix cores: a.qs / a.d / a.bsums = the expert-sorted image (a.ncol rows)
int cdna4_launch_mmq_ids(const cdna4_gemv_args &a, const int32_t *tile_expert, const int32_t *row_dst, int64_t w_expert_bytes, hipStream_t st);

// gemm_q_mfma.hip — fp16 MFMA prefill path.  xh = pair-interleaved fp16 activations [B][K].
struct cdna4_gemm_args {
    int type;
    const uint8_t *W; int64_t w_row_bytes;
    const void *xh; int64_t xh_row_elems;
    float *Y; int64_t y_row_elems;                  // Y[b * y_row_elems + m]
    int M, K, B;
    int variant;                                    // 0 = auto; see gemm_q_mfma.hip
    int splitk;                                     // 0 = auto
    cdna4_epilogue epi;                             // the MUL_MAT's element-wise tail (zeroed = none): applied IN THE STORE by the kernels for which
                                                    // cdna4_gemm_q_fuses_tail() holds; the others ignore it (the caller appends cdna4_launch_epilogue)
    // the fp32 activation rows (rows xf_row_elems apart), or nullptr: when given, a route for which cdna4_gemm_q_fuses_quantizer() holds writes the image `xh` ITSELF
    // (the one-launch step: k_gemm_kq_t64<.., FQ>); every other route ignores them and expects `xh` prepared (ggml_cdna4_prepare_act)
    const float *xf; int64_t xf_row_elems;
};
int cdna4_launch_gemm_q(const cdna4_gemm_args &a, hipStream_t st);
// does cdna4_launch_gemm_q(a) apply a.epi itself (bias / GELU / residual in the store of k_gemm_kq_t64)?  Then a residual may alias Y exactly.
bool cdna4_gemm_q_fuses_tail(const cdna4_gemm_args &a);
// does that route quantize a.xf itself (then the caller skips the activation quantizer's launch)?  The routing code itself, probed: no side effects.
bool cdna4_gemm_q_fuses_quantizer(const cdna4_gemm_args &a);
bool cdna4_gemm_t64_fuses_quantizer(const cdna4_gemm_args &a, int tm, int splitk);      // gemm_q_t64.hip: the launcher's own plan
int cdna4_gemm_q_route(const cdna4_gemm_args &a);              // the prefill kernel AUTO (or the given variant) would launch: ids in gemm_q_mfma.hip; no side effects
int cdna4_gemm_take_fault(bool clear);                 // gemm_q_mfma.hip: the fault code a waiting exchange reported (0: none); a fault also demotes the library to the shared mode
int cdna4_gemm_set_shared_device(int shared);          // gemm_q_mfma.hip: 1 = never choose a split-K exchange that spins on a co-resident partner; returns the old value
// gemm_q_t64.hip — grouped MUL_MAT_ID: a.B = rows of the expert-sorted activation image, a.Y rows indexed through row_dst
int cdna4_launch_gemm_t64_ids(const cdna4_gemm_args &a, const int32_t *tile_expert, const int32_t *row_dst, int64_t w_expert_bytes, hipStream_t st);
// gemm_q_sk.hip — the same as ONE stream-k launch of persistent work-groups over the plan of cdna4_launch_moe_sk_front (a.B = rows of the token-order image)
int cdna4_gemm_sk_spans();                                                                    // work-groups of the launch (= spans the planner cuts)
bool cdna4_gemm_sk_supported(int type, int64_t M, int64_t K, int64_t n_rows, int64_t ntile_cap);
int cdna4_launch_gemm_sk(const cdna4_gemm_args &a, const int32_t *tile_rec, const int32_t *wg_begin, int G, int64_t w_expert_bytes, hipStream_t st);
// the same for any of the five headline formats (gemm_q_mfma.hip): Q4_K on aligned rows -> k_gemm_kq_t64<.., IDS>, the others -> k_gemm_q<.., IDS>
bool cdna4_gemm_ids_supported(int type, int64_t K);
int cdna4_launch_gemm_ids(const cdna4_gemm_args &a, const int32_t *tile_expert, const int32_t *row_dst, int64_t w_expert_bytes, hipStream_t st);
// the activation image of the rows src_rows[0 .. img_rows) for the grouped MUL_MAT_ID (quantize_act.hip): Q8_K (K-quants) or Q8_0 quantization
int cdna4_launch_quantize_q8_0_gather(const float *x, int64_t x_row_stride, int64_t K, int64_t img_rows, const int32_t *src_rows, void *xh, hipStream_t st);
int cdna4_launch_quantize_q8_K_gather_i8(const float *x, int64_t x_row_stride, int64_t K, int64_t img_rows, const int32_t *src_rows, int8_t *qs, float *d, int16_t *bsums, hipStream_t st);
int cdna4_launch_quantize_q8_0_gather_i8(const float *x, int64_t x_row_stride, int64_t K, int64_t img_rows, const int32_t *src_rows, int8_t *qs, float *d, hipStream_t st);
bool cdna4_gemm_q_supported(int type, int64_t M, int64_t K, int64_t B);
// convert_w.hip: exact re-encodings Q5_0 -> Q8_0, Q3_K -> Q6_K (prefill GEMM of the source format = GEMM of the target format) and
// Q2_K -> [scale part | minimum part] as Q6_K with 2 K columns (kmul = 2: the activation image must hold x twice)
void *cdna4_gemm_scratch(size_t bytes, int kind);      // gemm_q_mfma.hip: per-device library scratch by kind (3 = re-encoded weights)
size_t cdna4_convert_weights_bytes(int type, int64_t M, int64_t K);
int cdna4_convert_weights_target(int type);
int cdna4_convert_weights_kmul(int type);
int cdna4_launch_convert_weights(int type, const uint8_t *W, int64_t w_row_bytes, int64_t M, int64_t K, uint8_t *out, hipStream_t st);

// resident kernel-native images of the re-encoded formats (gemm_q_mfma.hip): registry keyed by the weight pointer
struct ggml_cdna4_tensor;
extern "C" int cdna4_launch_norm_affine_q8_K(const ggml_cdna4_tensor *a, const ggml_cdna4_tensor *gain, const ggml_cdna4_tensor *shift, const ggml_cdna4_tensor *d, float eps, int rms, void *xh, void *stream, int kq);   // ops.hip (kq: the Q8_K image / 0: the Q8_0 image)
size_t cdna4_resident_image_row_bytes(int type, int64_t K);                 // 0: no image for this type / K
int cdna4_resident_build(int type, const uint8_t *W, int64_t w_row_bytes, int64_t M, int64_t K, uint8_t *out, hipStream_t st);
int cdna4_resident_register(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, const void *image);
int cdna4_resident_unregister(const void *W);
const uint8_t *cdna4_resident_lookup(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K);

extern void *cdna4_debug_trace;
uint64_t cdna4_scratch_generation();            // gemm_q_mfma.hip: bumped whenever the library (re)allocates device scratch
