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
 *   ggml_cdna4_quantize_q8_1    quantize_row_q8_1 (AVX2 body)          src/ggml-cpu/ggml-cpu-quants.c:1076-1119
 *                               (from_float of Q8_1 = vec_dot_type of Q4_1 / Q5_1: src/ggml-cpu/ggml-cpu.c:271-296)
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
    /* MUL_MAT / MUL_MAT_ID through the int8-dot GEMV units; above 8 activation rows MUL_MAT takes the MFMA GEMM of Q8_0 / Q6_K on an exact
     * re-encoding of the weights (Q5_0 / Q3_K: ggml_cdna4_convert_weights; Q2_K: scale part and minimum part as two Q6_K column blocks
     * against a doubled activation image, inside the library) */
    GGML_CDNA4_TYPE_Q5_0 = 6, GGML_CDNA4_TYPE_Q2_K = 10, GGML_CDNA4_TYPE_Q3_K = 11,
    /* the same.  Q4_1 / Q5_1: Q8_1 activations (the CPU's vec_dot_type, src/ggml-cpu/ggml-cpu.c:271-296), prefill as [d q | m 1] in Q8_0 against a
     * doubled activation image (inside the library, K a multiple of 128).  IQ4_NL: the codebook values are int8, so the weights re-encode
     * exactly as Q8_0 (ggml_cdna4_convert_weights) */
    GGML_CDNA4_TYPE_Q4_1 = 3, GGML_CDNA4_TYPE_Q5_1 = 7, GGML_CDNA4_TYPE_IQ4_NL = 20,
    /* the same; prefill as two Q6_K column blocks (codebook value = 4 h + l) against a doubled activation image, inside the library */
    GGML_CDNA4_TYPE_IQ4_XS = 23,
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
/* 1 (the DEFAULT since round 6): other work may hold CUs of this device while our kernels run (another process, another stream — ordinary for a ggml plug-in): the AUTO
 * routes never choose an exchange that WAITS for a co-resident partner work-group (the one-launch step's grid barrier, k_gemm_kq_t64's hand-off, k_gemm_r8's
 * reduce-scatter, the 128x128-tile kernels' hand-off) — small grids take the ticketed split (the last work-group to arrive sums; nobody waits) or no split, the
 * headline step is quantizer + GEMM: correct under any sharing, like the reference (whose stream-k partials are summed by a second launch, src/ggml-cuda/mmq.cuh:2796-2822).
 * 0 (or GGML_CDNA4_OWNED_DEVICE=1 in the environment): the caller OWNS the device — the waiting routes become eligible (measured worth 1-2 % at
 * [4096x4096]x[4096x512]; bench.py sets it and says so).  Returns the previous value. */
int          ggml_cdna4_set_shared_device(int shared);
/* Non-zero if, since the last clearing call, a launch on one of those waiting routes gave up waiting (1 grid barrier, 2 / 4 hand-off, 3 reduce-scatter): the device was not
 * exclusively ours after all, that launch's output holds NaN tiles.  Seeing a fault also switches the library to the shared mode for the rest of the process.  Every
 * ggml_cdna4_mul_mat* / _mul_mat_id call checks it first and returns -3 (ggml_cdna4_last_error says why) instead of launching — a wrong result is never handed back with
 * status 0; the plug-in's graph_compute returns GGML_STATUS_FAILED.  clear = 0 only looks. */
int          ggml_cdna4_device_fault(int clear);
/* test hook (tests/test_gpu_shared_device.py): n_workgroups work-groups that each hold lds_kb KB of LDS (64 threads) and spin until *release (host-visible memory, may be
 * null) becomes non-zero or max_ms milliseconds have passed — "another tenant" on the device.  Asynchronous on `stream`. */
int          ggml_cdna4_debug_occupy(int n_workgroups, int lds_kb, const int * release, int max_ms, void * stream);
/* Which route does ggml_cdna4_mul_mat(path = AUTO) take for a contiguous, 256-byte-aligned call of this shape on the current device?  Host logic only (no launch):
 *   1 one launch, activation quantizer inside the GEMV      2 quantize + GEMV      3 quantize + int8 matrix-core kernel (3..64 rows)
 *   10 quantize + k_gemm_kq_t64      12 quantize + k_gemm_r8      13 quantize + a 128x128-tile kernel      14 quantize + an older per-lane-load GEMM
 *   + 100: behind an exact re-encoding of the weights (ggml_cdna4_convert_weights);  0: not a supported call.
 * For hosts that want to know what a shape costs before they choose a batch size, and for the tests that pin the route table (profiles/r04/batch_sweep.txt). */
int          ggml_cdna4_mul_mat_route(int type, int64_t M, int64_t K, int64_t B);
/* the same for a concrete weight matrix: sees its alignment, row stride and a resident image registered for it (Q4_0 with an image: 10 / 12 instead of 13) */
int          ggml_cdna4_mul_mat_route_of(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, int64_t B);
int          ggml_cdna4_set_device(int device);
/* profiling hook (tools/microbench/gemm_bench only): a 64 KiB device buffer makes the 8-wave GEMM record per-phase
 * s_memtime stamps of its first work-group; NULL (default) selects the uninstrumented kernel */
void         ggml_cdna4_debug_trace(void * device_buffer);
/* Library scratch (split-K exchange areas, re-laid weights) is allocated lazily per device and grown on demand; a launch captured into
 * a HIP graph holds the address it was given.  This counter changes whenever any such allocation is made or moved — and whenever a resident image is registered or
 * unregistered (a captured launch holds "image found / not found" and the image's address): a host that replays captured launches compares it with the value at
 * capture time and re-captures on a mismatch (the plug-in does). */
uint64_t     ggml_cdna4_scratch_generation(void);

size_t ggml_cdna4_row_size(int type, int64_t k);          /* bytes of one row of k weights; 0 if unsupported */

/* bytes of scratch ggml_cdna4_mul_mat / _mul_mat_id need for (K, n activation rows) */
size_t ggml_cdna4_mul_mat_workspace_size(int type, int64_t K, int64_t n_act_rows);

/*
 * Y[b * y_row_stride + m] = sum_k W[m][k] * X[b * x_row_stride + k],  m < M, b < B.
 *   W: M rows of K block-quantized weights (type in {Q4_0,Q8_0,Q4_K,Q5_K,Q6_K} + {Q5_0,Q2_K,Q3_K,Q4_1,Q5_1,IQ4_NL,IQ4_XS}: GEMV units of their own,
 *      prefill through an exact re-encoding into Q8_0 / Q6_K), row stride w_row_bytes.
 *   X: f32, Y: f32; strides in ELEMENTS.  workspace: >= ggml_cdna4_mul_mat_workspace_size(type, K, B) bytes,
 *   256-byte aligned.  path: enum ggml_cdna4_path.  gemm_variant / splitk: 0 = auto (tuning knobs; the bit
 *   layout of gemm_variant is documented at launch_type() in ggml_amd/csrc/gemm_q_mfma.hip).
 *   Asynchronous on `stream`; the workspace must stay alive until the stream has passed the call.
 *   Which kernels run (auto):  B == 1 -> one launch, the activation quantizer fused into the int8-dot GEMV (the
 *   workspace is not touched);  2 <= B <= 8 -> quantize + GEMV;  B > 8 -> quantize (fp16 image) + MFMA GEMM.
 *   The GEMM may use library-owned device scratch (split-K exchange buffers; a per-call 16-byte-aligned re-layout
 *   of Q4_0 / Q8_0 / Q6_K weights at B > 64): one scratch set per device, so GEMM calls on ONE device must be issued
 *   from one stream at a time (what a ggml backend does anyway: a ggml_backend_t is a single-stream object).  The
 *   weights themselves are only read.  Results are deterministic for fixed (shape, gemm_variant, splitk).
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
 *   b  : f32 [n_tok][n_b][K]   (n_used % n_b == 0; slot u reads row u % n_b)     strides in elements
 *   ids: i32 [n_tok][n_used]   (device memory; never copied to the host)        stride in elements
 *   dst: f32 [n_tok][n_used][M]
 * n_tok == 1 (decode) runs as ONE launch with the activation quantizer inside the GEMV (workspace untouched); a handful of
 * tokens quantize the activations into `workspace` and run the int8-dot GEMV per (token, slot) column; prefill-sized batches
 * (Q4_K, n_tok * n_used > 32, workspace >= ggml_cdna4_mul_mat_id_workspace_size) are GROUPED BY EXPERT on the device — a
 * counting sort of the ids, no host sync — and run as one MFMA GEMM launch over the (expert, activation tile) table, the way
 * ggml_compute_forward_mul_mat_id groups rows on the host (ggml-cpu.c:7648-7781).  Expert ids out of [0, n_expert) leave
 * their slot unwritten.
 */
size_t ggml_cdna4_mul_mat_id_workspace_size(int type, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok);
int ggml_cdna4_mul_mat_id(int type, const void * as, int64_t w_row_bytes, int64_t w_expert_bytes,
                          const float * b, int64_t b_row_stride, int64_t b_tok_stride,
                          const int32_t * ids, int64_t ids_tok_stride,
                          float * dst, int64_t dst_row_stride, int64_t dst_tok_stride,
                          int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok,
                          void * workspace, size_t workspace_bytes, void * stream);

/* MUL_MAT_IDs of ONE (b, ids): the up- and gate-projection expert stacks of a mixture-of-experts layer multiply the same activations routed by the same ids (the reference
 * evaluates them as two independent nodes, each sorting and quantizing again: ggml-cpu.c:7609-7784).  ggml_cdna4_mul_mat_id_front_key != 0: a ggml_cdna4_mul_mat_id of this call
 * takes the work-queue form and leaves its FRONT — sorted ids, tile records, spans, quantized activations — in the workspace; ggml_cdna4_mul_mat_id_prepared (same arguments,
 * other expert weights of the same type / M / K; same b, ids, strides and counts; nothing written to b, ids or the workspace in between) multiplies that front again: ONE launch
 * instead of two, bit-identical to the full call.  Returns -2 (nothing launched) where the call has no such form. */
uint32_t ggml_cdna4_mul_mat_id_front_key(int type, const void * as, int64_t w_row_bytes, int64_t w_expert_bytes, int64_t M, int64_t K,
                                         int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok, size_t workspace_bytes);
int ggml_cdna4_mul_mat_id_prepared(int type, const void * as, int64_t w_row_bytes, int64_t w_expert_bytes,
                                   const float * b, int64_t b_row_stride, int64_t b_tok_stride,
                                   const int32_t * ids, int64_t ids_tok_stride,
                                   float * dst, int64_t dst_row_stride, int64_t dst_tok_stride,
                                   int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok,
                                   void * workspace, size_t workspace_bytes, void * stream);

/* MUL_MAT with its element-wise tail: Y[b][m] = (((W.X)[b][m] + bias[m]) -> GELU if act == 1) + residual[b][m]; bias / residual may be NULL.
 * What the gpt-2 graphs do in three nodes after every projection — MUL_MAT, ADD(bias), then GELU or ADD(residual)
 * (/root/reference/examples/gpt-2/main-backend.cpp:515-521, 595-600, 656-666, 690-698).  Each step is the same separate fp32 operation the
 * stand-alone ops perform (GELU = the CPU's fp16 look-up-table semantics, ggml-cpu.c:1759-1774), so the result is BIT-IDENTICAL to
 * ggml_cdna4_mul_mat -> ggml_cdna4_op_binary(ADD) -> ggml_cdna4_op_unary(GELU) / ggml_cdna4_op_binary(ADD).  Decode-sized batches apply
 * the tail in the GEMV's store (one launch); the MFMA GEMM path appends ONE element-wise launch for the whole tail. */
int ggml_cdna4_mul_mat_fused(int type, const void * W, int64_t w_row_bytes, const float * X, int64_t x_row_stride, float * Y, int64_t y_row_stride,
                             int64_t M, int64_t K, int64_t B, const float * bias, int act, const float * residual, int64_t residual_row_stride,
                             void * workspace, size_t workspace_bytes, void * stream);
/* n (1 .. 4) MUL_MATs of ONE activation row in ONE launch — Y[i][m] = sum_k W[i][m][k] X[k] (+ bias[i][m]) for matrices of one type and K: wq / wk / wv and
 * w_gate / w_up of a decoded token read the same src1 (examples/gpt-2/main-backend.cpp:476-521 has the fused c_attn; llama-style graphs have them apart).  Bit-identical
 * to n ggml_cdna4_mul_mat / _mul_mat_fused calls with B = 1.  Returns -2 (nothing launched) where the call has no grouped form: the caller then issues the separate calls. */
int ggml_cdna4_mul_mat_group(int type, int n, const void * const * W, const int64_t * w_row_bytes, const int64_t * M, float * const * Y, const float * const * bias,
                             const float * X, int64_t K, void * stream);

/* `residual` may alias Y EXACTLY (same pointer and row stride: an in-place ADD) only when the tail is applied by the store that produces the
 * element; this says whether that holds for a call of this shape (1) or whether residual and Y must not overlap at all (0: the product is
 * written first and the tail is a pass over Y — an aliased residual would already be overwritten).  ggml_cdna4_mul_mat_fused returns an error
 * for an overlap it cannot honour; partial overlaps are always an error. */
int ggml_cdna4_mul_mat_fused_residual_may_alias(int type, int64_t M, int64_t K, int64_t B);

/* Hand-off of quantized activations between MUL_MATs that read the SAME src1 (wq / wk / wv, w_gate / w_up): the CPU backend quantizes src1 once per node
 * (src/ggml-cpu/ggml-cpu.c:7490-7509), and so does every ggml_cdna4_mul_mat call.  ggml_cdna4_act_image_key says which image a call of this shape leaves in its
 * workspace (0: none that can be reused).  If the previous call on the SAME workspace had the same non-zero key, X, x_row_stride, K and B, and neither X nor the
 * workspace has been written since, the next product may be ggml_cdna4_mul_mat_prepared(path = GGML_CDNA4_PATH_AUTO) or its fused twin below: the same kernel on
 * the same image — bit-identical to ggml_cdna4_mul_mat[_fused], one launch fewer.  The plug-in's graph walk does exactly this (GGML_CDNA4_NO_ACT_SHARE=1: off). */
uint32_t ggml_cdna4_act_image_key(int type, int64_t M, int64_t K, int64_t B);
/* the same for a concrete weight matrix (like ggml_cdna4_mul_mat_route_of): a few-row call on rows that are not 16-byte aligned leaves no int8 image (it may quantize
 * inside a one-launch GEMV and never touch the workspace) — hosts that record the key after a call must ask with the pointers of THAT call */
uint32_t ggml_cdna4_act_image_key_of(int type, const void * W, int64_t w_row_bytes, int64_t M, int64_t K, int64_t B);
int ggml_cdna4_mul_mat_prepared_fused(int type, const void * W, int64_t w_row_bytes, float * Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B,
                                      const float * bias, int act, const float * residual, int64_t residual_row_stride,
                                      const void * workspace, size_t workspace_bytes, void * stream);

/* Activation quantizers (bit-exact with the reference); outputs may be NULL to skip them.
 *   qs  int8  [B][K]      d  f32 [B][K/256 | K/32]      bsums int16 [B][K/16] (Q8_K only)
 *   xh  fp16  B*K halves = fp16(d*q) in the MFMA path's private layout: element (b,k) at ((k/128)*B + b)*128 + k%128
 *       (k-panel-major), and within every 4 consecutive k the order (k0,k2,k1,k3) (pair-interleaved) */
int ggml_cdna4_quantize_q8_K(const float * x, int64_t x_row_stride, int64_t K, int64_t B,
                             int8_t * qs, float * d, int16_t * bsums, void * xh, void * stream);
int ggml_cdna4_quantize_q8_0(const float * x, int64_t x_row_stride, int64_t K, int64_t B,
                             int8_t * qs, float * d, void * xh, int ref_rounding, void * stream);
/* Q8_1 (activations of Q4_1 / Q5_1 weights): the Q8_0 block of the AVX2 body plus s f32 [B][K/32] holding block_q8_1.s =
 * fp16(d * sum of the block's 32 quants) with d still in fp32 (two roundings: fp32 product, then fp16), widened to fp32 like d */
int ggml_cdna4_quantize_q8_1(const float * x, int64_t x_row_stride, int64_t K, int64_t B,
                             int8_t * qs, float * d, float * s, void * xh, void * stream);


/* ---------------------------------------------------------------------------------------------------------
 * Supporting ops (plain HIP kernels; the reference's counterparts are the ggml_compute_forward_* functions of
 * src/ggml-cpu/ggml-cpu.c cited per entry).  Tensors are described without ggml types: data pointer, element
 * type (enum ggml_cdna4_type / GGML_CDNA4_TYPE_I32 = 26), ne[4] elements and nb[4] BYTE strides, in ggml's
 * dimension order (ne[0] fastest) — exactly the fields of struct ggml_tensor (include/ggml.h:576-608).
 * --------------------------------------------------------------------------------------------------------- */
#define GGML_CDNA4_TYPE_I32 26

typedef struct ggml_cdna4_tensor {
    void *  data;
    int32_t type;
    int32_t reserved;
    int64_t ne[4];
    int64_t nb[4];
} ggml_cdna4_tensor;

enum ggml_cdna4_binary_op { GGML_CDNA4_ADD = 0, GGML_CDNA4_SUB = 1, GGML_CDNA4_MUL = 2, GGML_CDNA4_DIV = 3 };
enum ggml_cdna4_unary_op  { GGML_CDNA4_GELU = 0, GGML_CDNA4_GELU_QUICK = 1, GGML_CDNA4_SILU = 2, GGML_CDNA4_RELU = 3, GGML_CDNA4_TANH = 4 };

/* dst = src0 (op) broadcast(src1), all F32 — ggml_compute_forward_add/sub/mul/div, ggml-cpu.c:4052-5260 */
int ggml_cdna4_op_binary(int op, const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * src1, const ggml_cdna4_tensor * dst, void * stream);
/* dst = src0 * scale — ggml_compute_forward_scale, ggml-cpu.c:8047-8100 */
/* dst[i] = parts[0][i] + parts[1][i] + .. + parts[n-1][i], in that order (1 <= n <= 16; 16-byte aligned; dst may be parts[0]).
 * The deterministic reduction of a K-split MUL_MAT's partial outputs — what ggml_cuda_op_mul_mat leaves to its caller for a row split and what
 * the north star's "all-reduce on the activations" is when the shards share a device (the plug-in uses RCCL across devices: ggml_cdna4_split.cpp). */
int ggml_cdna4_sum_partials(float * dst, const float * const * parts, int n, int64_t count, void * stream);
int ggml_cdna4_op_scale(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, float scale, void * stream);
/* LayerNorm without affine / RMSNorm over ne[0] — ggml-cpu.c:6929-6978, 7000-7046 */
int ggml_cdna4_op_norm(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, float eps, int rms, void * stream);
/* NORM / RMS_NORM followed by the MUL(gain) and ADD(shift) of a LayerNorm (either may be NULL; F32 vectors of ne[0] elements):
 * bit-identical to op_norm -> op_binary(MUL) -> op_binary(ADD) (gpt-2: main-backend.cpp:476-488) */
int ggml_cdna4_op_norm_affine(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * gain, const ggml_cdna4_tensor * shift, const ggml_cdna4_tensor * dst,
                              float eps, int rms, void * stream);
/* NORM / RMS_NORM [* gain] [+ shift] (ggml_cdna4_op_norm_affine, same bits in dst) that ALSO leaves in `workspace` the activation image a following
 * ggml_cdna4_mul_mat(type, .., X = dst) would build, where that call's ggml_cdna4_act_image_key is the K-quants' fp16 GEMM image (== 19): the MUL_MATs that read dst
 * then run ggml_cdna4_mul_mat_prepared[_fused] and no quantizer launch is paid for them.  dst: contiguous F32 rows of 256 .. 8192 values (whole superblocks). */
int ggml_cdna4_op_norm_affine_q8_K(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * gain, const ggml_cdna4_tensor * shift, const ggml_cdna4_tensor * dst,
                                   float eps, int rms, int type, void * workspace, size_t workspace_bytes, void * stream);
/* the same for the 32-block formats whose activations are Q8_0 (Q4_0 / Q8_0 / Q5_0 / IQ4_NL: ggml_cdna4_act_image_key == 17); rows of whole 32-value blocks, up to 8192 */
int ggml_cdna4_op_norm_affine_q8_0(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * gain, const ggml_cdna4_tensor * shift, const ggml_cdna4_tensor * dst,
                                   float eps, int rms, int type, void * workspace, size_t workspace_bytes, void * stream);
/* softmax(src0*scale + slope*mask) over ne[0]; mask (F32 or F16, [ne0, ne1]) may be NULL; ALiBi slopes from
 * max_bias as ggml-cpu.c:8848-8944 */
int ggml_cdna4_op_soft_max(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * mask, const ggml_cdna4_tensor * dst, float scale, float max_bias, void * stream);
/* soft_max with the SCALE (use_pre_scale != 0: x * pre_scale first) and DIAG_MASK_INF (diag_n_past >= 0) nodes that precede it in
 * causal attention (gpt-2: main-backend.cpp:586-596) folded in, bit-identical to op_scale -> op_diag_mask_inf -> op_soft_max */
int ggml_cdna4_op_soft_max_ext(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * mask, const ggml_cdna4_tensor * dst, float scale, float max_bias,
                               int use_pre_scale, float pre_scale, int diag_n_past, void * stream);
/* dst = src0 with dst[.., j, i] = -inf for i > n_past + j — ggml-cpu.c:8760-8830 */
int ggml_cdna4_op_diag_mask_inf(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, int n_past, void * stream);
/* element-wise activations; GELU reproduces the CPU's fp16 lookup-table semantics (ggml-cpu.c:1759-1774) */
int ggml_cdna4_op_unary(int op, const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, void * stream);
/* dst[.., i10, :] = to_float(src0 row ids[i10, i11, i12]) — ggml_compute_forward_get_rows, ggml-cpu.c:8353-8560;
 * src0 in {F32, F16} or any of the twelve block formats of enum ggml_cdna4_type */
int ggml_cdna4_op_get_rows(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * ids, const ggml_cdna4_tensor * dst, void * stream);
/* CPY / DUP / CONT: copy with conversion between tensors of equal element count — ggml_compute_forward_dup,
 * ggml-cpu.c:2860-4050.  Pairs: {F32,F16}->{F32,F16}; F32->{Q8_0,Q4_0,Q4_1,Q5_0,Q5_1} (from_float of the type); any block format of enum ggml_cdna4_type -> F32.
 * q8_0_ref_rounding: 0 = the CPU backend's from_float (AVX2 body), 1 = quantize_row_q8_0_ref. */
int ggml_cdna4_op_cpy(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, int q8_0_ref_rounding, void * stream);
/* MUL_MAT with F32 or F16 weights and F32 activations, any strides / batch broadcast —
 * ggml_compute_forward_mul_mat with vec_dot_f32 / vec_dot_f16 (ggml-cpu.c:7428-7605) */
int ggml_cdna4_op_mul_mat_f(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * src1, const ggml_cdna4_tensor * dst, void * stream);

/* ---- reference-ORDER forms (ggml_amd/csrc/exact.hip), opt-in through the plug-in's GGML_CDNA4_EXACT=1: the ops of a gpt-2 graph whose fp32 summation order
 * differs from the CPU backend's by default, computed in the order of the x86-64-v3 (AVX2 + FMA) build of the reference, so that a whole graph reproduces the
 * CPU backend's logits bit for bit (BASELINE.json configs[3]).  Verification mode: order costs speed.
 * MUL_MAT, Q4_0 / Q8_0 weights: quantize_row_q8_0 (AVX2 body) + ggml_vec_dot_q4_0_q8_0 / _q8_0_q8_0 with eight lane accumulators and hsum_float_8
 * (src/ggml-cpu/ggml-cpu-quants.c:778-815, 2005-2028, 3520-3536, 49-55); Q4_K / Q5_K / Q6_K weights (K a multiple of 256): quantize_row_q8_K +
 * the AVX2 bodies of ggml_vec_dot_q4_K_q8_K / _q5_K_q8_K / _q6_K_q8_K (:5712-5775, 6283-6364, 6941-7018: eight lane accumulators, the mins in Q4_K's four-lane
 * acc_m / Q5_K's scalar summs); workspace >= ggml_cdna4_mul_mat_exact_workspace_size, 256-byte aligned */
int    ggml_cdna4_mul_mat_exact_supported(int type, int64_t K);
size_t ggml_cdna4_mul_mat_exact_workspace_size(int type, int64_t K, int64_t B);
int    ggml_cdna4_mul_mat_exact(int type, const void * W, int64_t w_row_bytes, const float * X, int64_t x_row_stride, float * Y, int64_t y_row_stride,
                                int64_t M, int64_t K, int64_t B, void * workspace, size_t workspace_bytes, void * stream);
/* MUL_MAT F32 x F32 as ggml_vec_dot_f32 (src/ggml-cpu/ggml-cpu.c:1346-1377, GGML_F32x8_REDUCE :670-688, gcc's leftover loop) */
int    ggml_cdna4_op_mul_mat_f_exact(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * src1, const ggml_cdna4_tensor * dst, void * stream);
/* NORM with sequential double sums (ggml_compute_forward_norm_f32, ggml-cpu.c:6929-6978) */
int    ggml_cdna4_op_norm_exact(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, float eps, void * stream);
/* RMS_NORM with the sequential double sum of fp32 squares (ggml_compute_forward_rms_norm_f32, ggml-cpu.c:7000-7046) */
int    ggml_cdna4_op_rms_norm_exact(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, float eps, void * stream);
/* SILU on contiguous F32 rows: ggml_v_silu per chunk of 8 of a row, x / (1 + expf(-x)) with glibc's expf on the row's tail (ggml_vec_silu_f32, ggml-cpu.c:2017-2039, 1952-1959) */
int    ggml_cdna4_op_silu_exact(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, void * stream);
/* SOFT_MAX without mask / ALiBi on contiguous rows: ggml_v_expf per chunk of 8, the chunk sums in the AVX2 shuffle order accumulated in double, glibc's expf
 * on the tail (ggml_compute_forward_soft_max_f32, ggml-cpu.c:8848-8944, 2041-2092, 1912-1949) */
int    ggml_cdna4_op_soft_max_exact(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * dst, float scale, void * stream);
/* rotary embedding, modes NORMAL (0) and NEOX (2), with freq_base/freq_scale/ext_factor(yarn)/attn_factor and
 * optional freq_factors — ggml_compute_forward_rope_f32, ggml-cpu.c:9255-9625 */
int ggml_cdna4_op_rope(const ggml_cdna4_tensor * src0, const ggml_cdna4_tensor * pos, const ggml_cdna4_tensor * freq_factors, const ggml_cdna4_tensor * dst,
                       int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor,
                       float beta_fast, float beta_slow, void * stream);
/* GGML_OP_FLASH_ATTN_EXT — ggml_compute_forward_flash_attn_ext_f16, src/ggml-cpu/ggml-cpu.c:10805-11016 (ggml-cuda: fattn*.cu).
 * q F32 [head_size, n_q, n_head, batch] (any row strides), k / v F16 [head_size, n_kv, n_head_kv, batch_kv] (rows 16-byte aligned; heads and
 * batches broadcast as q's over k's), mask F16 [n_kv, >= n_q] or NULL, dst F32 contiguous [head_size, n_head, n_q, batch].
 * scale / max_bias (ALiBi) / logit_softcap as in op_params 0..2.  fp16 operands on the matrix cores, fp32 softmax statistics and
 * accumulation.  k / v may also be BF16 (type 30) or block-quantized (Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0: a quantized KV cache; rows contiguous, any
 * strides).  Decode-sized calls (the key-split kernel: up to 32 query rows, or a grid of 128-row tiles that would not fill the chip) read a
 * Q8_0 / Q4_0 / BF16 cache DIRECTLY — fp16(to_float(element)) in the operand loads, k and v of the same type, rows 4-byte (BF16: 16-byte) aligned;
 * everything else is written out as fp16 into library scratch first (one pass), then the same kernels run: bit-identical either way.  Head sizes
 * other than 64 / 128 / 256 (80, 96, 112, ...; up to 256) run zero-padded to the next of them through padded copies of q / k / v and of the result.
 * _supported: head size 1..256 with F16 / BF16 k / v, a multiple of 32 with quantized k / v. */
int ggml_cdna4_op_flash_attn_ext_supported(int64_t head_size, int kv_type);
int ggml_cdna4_op_flash_attn_ext(const ggml_cdna4_tensor * q, const ggml_cdna4_tensor * k, const ggml_cdna4_tensor * v, const ggml_cdna4_tensor * mask,
                                 const ggml_cdna4_tensor * dst, float scale, float max_bias, float logit_softcap, void * stream);
/* to_float of a quantized row buffer: y[k] f32 <- x (type) — dequantize_row_*, src/ggml-quants.c */
int ggml_cdna4_dequantize_row(int type, const void * x, float * y, int64_t k, void * stream);

/* Exact re-encoding of a weight matrix into the format whose MFMA prefill GEMM it shares: Q5_0 -> Q8_0 (q8 = q5 - 16, same d),
 * IQ4_NL -> Q8_0 (q8 = kvalues_iq4nl[code], same d) and Q3_K -> Q6_K (q6 = q3 + 28, int8 scale = 6-bit scale - 32, same d).  dequantize_row of the result equals dequantize_row of the source bit
 * for bit (src/ggml-quants.c:295-319 vs 349-363, 1056-1104 vs 1690-1719), and both members of a pair use the same activation format on the
 * CPU (type_traits_cpu[].vec_dot_type, src/ggml-cpu/ggml-cpu.c:277-341).  ggml_cdna4_mul_mat does this per call into library scratch; a
 * host that keeps prefill weights resident can convert once and call ggml_cdna4_mul_mat with the target type instead.
 * _target: the target type id or -1; _size: bytes of the result (rows contiguous) or 0. */
int    ggml_cdna4_convert_weights_target(int type);
size_t ggml_cdna4_convert_weights_size(int type, int64_t M, int64_t K);
int    ggml_cdna4_convert_weights(int type, const void * W, int64_t w_row_bytes, int64_t M, int64_t K, void * out, void * stream);

/* Resident kernel-native images (round 5).  ggml_cdna4_mul_mat re-encodes Q5_0 / IQ4_NL / Q4_1 / Q5_1 (-> Q8_0) and Q3_K / Q2_K / IQ4_XS (-> Q6_K; the two-part forms of
 * Q2_K / Q4_1 / Q5_1 / IQ4_XS included) into library scratch on EVERY prefill call.  A host that keeps a weight matrix resident registers an image instead: built once into
 * memory the host owns (_size bytes, 256-byte aligned, same device), with verify != 0 built TWICE and compared byte for byte — a re-encoding that is not bit-stable on this
 * device is refused here, at load time — and from then on every ggml_cdna4_mul_mat / _mul_mat_fused / _mul_mat_prepared call whose W is that matrix (or a row slice of it,
 * same type, K and row stride) finds the image by the pointer and launches no conversion.  The original bytes stay what they are: decode-sized calls read them as before
 * (their integer-dot units are the CPU's own arithmetic on the source format), get_tensor needs no inverse.  Synchronous: returns when the image is complete.
 * What the reference does with a repacking buffer type: src/ggml-cpu/ggml-cpu-aarch64.cpp:4144-4172 (repack at set_tensor), src/ggml-cpu/ggml-cpu.cpp:581-582.
 * _size: 0 = the type has no image.  _unregister before the weights or the image are freed.  _lookup: 1 and *image if a call with these arguments would use one. */
size_t ggml_cdna4_resident_image_size(int type, int64_t M, int64_t K);
int    ggml_cdna4_resident_image_register(int type, const void * W, int64_t w_row_bytes, int64_t M, int64_t K, void * image, int verify, void * stream);
int    ggml_cdna4_resident_image_unregister(const void * W);
int    ggml_cdna4_resident_image_lookup(int type, const void * W, int64_t w_row_bytes, int64_t M, int64_t K, const void ** image);

#ifdef __cplusplus
}
#endif
