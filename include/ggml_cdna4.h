/*
 * ggml_cdna4.h — C-ABI of the MI355X (gfx950 / CDNA4) kernel library `libcdna4_kernels.so`.
 *
 * This is the drop-in boundary BELOW ggml's backend plug-in API: plain pointers and sizes, no ggml, torch or
 * C++ types.  Every pointer is a DEVICE pointer unless it says "host"; `stream` is a hipStream_t (NULL = the
 * default stream).  All entry points return 0 (== GGML_STATUS_SUCCESS, include/ggml.h:320-325) or a negative
 * status (-1 == GGML_STATUS_FAILED, -2 == GGML_STATUS_ALLOC_FAILED); ggml_cdna4_last_error() describes the
 * failure.  Nothing here falls back to the CPU: a missing GPU or a failed launch is an error.
 *
 * The ggml plug-in `libggml-cdna4.so` (ggml_amd/csrc/backend/, entry point `ggml_backend_init`,
 * src/ggml-backend-impl.h:215) is a thin C++ layer that maps ggml_tensor strides onto these calls; a foreign
 * host (Go/Rust/Python ctypes) binds exactly this header.
 *
 * Reference interfaces each entry point replaces (paths relative to the reference tree):
 *   ggml_cdna4_mul_mat          ggml_compute_forward_mul_mat          src/ggml-cpu/ggml-cpu.c:7428-7605
 *                               (ggml-cuda: ggml_cuda_mul_mat          src/ggml-cuda/ggml-cuda.cu:1844-1905)
 *   ggml_cdna4_mul_mat_id       ggml_compute_forward_mul_mat_id       src/ggml-cpu/ggml-cpu.c:7609-7784
 *   ggml_cdna4_quantize_q8_K    quantize_row_q8_K (from_float of Q8_K) src/ggml-quants.c:2479-2516
 *   ggml_cdna4_quantize_q8_0    quantize_row_q8_0 (AVX2 body / _ref)   src/ggml-cpu/ggml-cpu-quants.c:778-815,
 *                                                                       src/ggml-quants.c:194-217
 *   ggml_cdna4_dequantize_row   type_traits[].to_float                 src/ggml-quants.c:255,349,1280,1482,1690
 *   ggml_cdna4_row_size         ggml_row_size                          src/ggml.c:1176-1179
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_CDNA4_API_VERSION 1

/* weight / tensor types: numerically identical to enum ggml_type (include/ggml.h:351-390) */
enum ggml_cdna4_type {
    GGML_CDNA4_TYPE_F32 = 0, GGML_CDNA4_TYPE_F16 = 1, GGML_CDNA4_TYPE_Q4_0 = 2, GGML_CDNA4_TYPE_Q8_0 = 8,
    GGML_CDNA4_TYPE_Q4_K = 12, GGML_CDNA4_TYPE_Q5_K = 13, GGML_CDNA4_TYPE_Q6_K = 14,
};

/* which kernel family ggml_cdna4_mul_mat uses */
enum ggml_cdna4_path {
    GGML_CDNA4_PATH_AUTO = 0,   /* B <= 8: int8-dot GEMV, else fp16-MFMA GEMM when the shape allows */
    GGML_CDNA4_PATH_GEMV = 1,   /* wave-reduction v_dot4_i32_i8 path (exact integer block sums) */
    GGML_CDNA4_PATH_GEMM = 2,   /* dequant -> fp16 MFMA path */
};

int          ggml_cdna4_api_version(void);
const char * ggml_cdna4_last_error(void);                 /* thread-local, never NULL */
int          ggml_cdna4_device_count(void);               /* number of visible HIP devices (0 if none) */
int          ggml_cdna4_set_device(int device);

size_t ggml_cdna4_row_size(int type, int64_t k);          /* bytes of one row of k weights; 0 if unsupported */

/* bytes of scratch ggml_cdna4_mul_mat / _mul_mat_id need for (K, n activation rows) */
size_t ggml_cdna4_mul_mat_workspace_size(int type, int64_t K, int64_t n_act_rows);

/*
 * Y[b * y_row_stride + m] = sum_k W[m][k] * X[b * x_row_stride + k],  m < M, b < B.
 *   W: M rows of K block-quantized weights (type in {Q4_0,Q8_0,Q4_K,Q5_K,Q6_K}), row stride w_row_bytes.
 *   X: f32, Y: f32; strides in ELEMENTS.  workspace: >= ggml_cdna4_mul_mat_workspace_size(type, K, B) bytes,
 *   256-byte aligned.  path: enum ggml_cdna4_path.  gemm_variant / splitk: 0 = auto (tuning knobs).
 */
int ggml_cdna4_mul_mat(int type, const void * W, int64_t w_row_bytes,
                       const float * X, int64_t x_row_stride,
                       float * Y, int64_t y_row_stride,
                       int64_t M, int64_t K, int64_t B,
                       void * workspace, size_t workspace_bytes,
                       int path, int gemm_variant, int splitk, void * stream);

/* Same, with activations already prepared by ggml_cdna4_prepare_act (weights-stationary serving loops and
 * the benchmark time this call: the hot kernel only). */
int ggml_cdna4_prepare_act(int type, const float * X, int64_t x_row_stride, int64_t K, int64_t B,
                           void * workspace, size_t workspace_bytes, int path, void * stream);
int ggml_cdna4_mul_mat_prepared(int type, const void * W, int64_t w_row_bytes,
                                float * Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B,
                                const void * workspace, size_t workspace_bytes,
                                int path, int gemm_variant, int splitk, void * stream);

/*
 * MUL_MAT_ID (mixture-of-experts routing), include/ggml.h ggml_mul_mat_id:
 *   as : n_expert matrices of M rows x K weights, expert stride w_expert_bytes
 *   b  : f32 [n_tok][n_b][K]   (n_b == n_used or 1; slot u reads row u % n_b)   strides in elements
 *   ids: i32 [n_tok][n_used]   (device memory; never copied to the host)        stride in elements
 *   dst: f32 [n_tok][n_used][M]
 */
int ggml_cdna4_mul_mat_id(int type, const void * as, int64_t w_row_bytes, int64_t w_expert_bytes,
                          const float * b, int64_t b_row_stride, int64_t b_tok_stride,
                          const int32_t * ids, int64_t ids_tok_stride,
                          float * dst, int64_t dst_row_stride, int64_t dst_tok_stride,
                          int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok,
                          void * workspace, size_t workspace_bytes, void * stream);

/* Activation quantizers (bit-exact with the reference); outputs may be NULL to skip them.
 *   qs  int8  [B][K]      d  f32 [B][K/256 | K/32]      bsums int16 [B][K/16] (Q8_K only)
 *   xh  fp16  [B][K]  = fp16(d*q), stored pair-interleaved (k0,k2,k1,k3 within every 4) for the MFMA path */
int ggml_cdna4_quantize_q8_K(const float * x, int64_t x_row_stride, int64_t K, int64_t B,
                             int8_t * qs, float * d, int16_t * bsums, void * xh, void * stream);
int ggml_cdna4_quantize_q8_0(const float * x, int64_t x_row_stride, int64_t K, int64_t B,
                             int8_t * qs, float * d, void * xh, int ref_rounding, void * stream);

#ifdef __cplusplus
}
#endif
