// capi.hip — the C-ABI of libcdna4_kernels.so (declared in include/ggml_cdna4.h).
#include "../../include/ggml_cdna4.h"
#include "cdna4_common.h"
#include "cdna4_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

int cdna4_set_error(hipError_t e, const char *file, int line) {
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
    return -1;
}
int cdna4_set_error_msg(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// Q5_0 / Q3_K / Q2_K: int8-dot GEMV units of gemv_q.hip for up to 8 activation rows, above that the Q8_0 / Q6_K MFMA GEMM on an exact
// re-encoding of the weights (convert_w.hip; Q2_K as [scale part | minimum part] against a doubled activation image)
static inline bool is_kq(int t) { return t == CDNA4_Q4_K || t == CDNA4_Q5_K || t == CDNA4_Q6_K || t == CDNA4_Q2_K || t == CDNA4_Q3_K || t == CDNA4_IQ4_XS; }
// Q4_1 / Q5_1 (Q8_1 activations: s in the place of the bsums) and IQ4_NL: GEMV units; above 8 rows the Q8_0 GEMM on the re-encoding
// (Q4_1 / Q5_1 as [d q | m 1] against a doubled activation image, like Q2_K)
static inline bool is_q(int t) { return is_kq(t) || t == CDNA4_Q4_0 || t == CDNA4_Q8_0 || t == CDNA4_Q5_0 || t == CDNA4_Q4_1 || t == CDNA4_Q5_1 || t == CDNA4_IQ4_NL; }

// workspace carve: [qs int8 B*K][d f32 B*K/qka][bsums i16 B*K/16][xh f16 B*K]
struct ws_view { int8_t *qs; float *d; int16_t *bsums; void *xh; size_t total; };
static ws_view carve(int type, int64_t K, int64_t B, void *base) {
    ws_view v; uint8_t *p = (uint8_t *)base; size_t off = 0;
    const int64_t qka = is_kq(type) ? 256 : 32;
    v.qs = (int8_t *)(p + off); off += align256((size_t)(B * K));
    v.d = (float *)(p + off); off += align256((size_t)(B * (K / qka)) * 4);
    v.bsums = (int16_t *)(p + off); off += align256((size_t)(B * (K / 16)) * 2);
    const int64_t kh = K * cdna4_convert_weights_kmul(type);       // Q2_K's GEMM reads the image twice in a row (convert_w.hip)
    v.xh = (void *)(p + off); off += align256((size_t)(B * ((kh + 127) / 128 * 128)) * 2) + 32768;    // whole 128-k panels; + slack: k_gemm_kq_t64 reads (and discards) up to 127 rows past a ragged last activation tile
    v.total = off;
    return v;
}

// workspace of the grouped MUL_MAT_ID path: [img_src i32 R][img_dst i32 R][tile_expert | tile_order | tile_nfrag: 3 x i32 R/128][xh f16 R x K (+ slack)], R = image rows =
// an upper bound of sum_e ceil(cnt_e / 128) * 128
struct moe_view { int32_t *img_src, *img_dst, *tile_expert; void *xh; int64_t img_rows; size_t total; };
static moe_view moe_carve(int64_t K, int64_t n_expert, int64_t n_used, int64_t n_tok, void *base) {
    moe_view v; uint8_t *p = (uint8_t *)base; size_t off = 0;
    const int64_t n_pairs = n_tok * n_used;
    v.img_rows = (n_pairs + 127) / 128 * 128 + 128 * n_expert;
    v.img_src = (int32_t *)(p + off); off += align256((size_t)v.img_rows * 4);
    v.img_dst = (int32_t *)(p + off); off += align256((size_t)v.img_rows * 4);
    v.tile_expert = (int32_t *)(p + off); off += align256((size_t)(v.img_rows / 128) * 4 * 3);     // [tile_expert | tile_order | tile_nfrag] (k_moe_plan writes all three)
    v.xh = (void *)(p + off); off += align256((size_t)v.img_rows * K * 2) + 32768;
    v.total = off;
    return v;
}

// the stream-k form of the grouped MUL_MAT_ID (round 6; quantize_act.hip: cdna4_launch_moe_sk_front, gemm_q_sk.hip): tile records, the spans, and the fp16 image of the
// n_tok * n_b activation rows in TOKEN order (the GEMM gathers a tile's rows itself: no per-pair copies, no padding rows)
struct moe_sk_view { int32_t *tile_rec, *wg_begin; void *xh; int64_t n_rows, ntile_cap; size_t total; };
static moe_sk_view moe_sk_carve(int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok, void *base) {
    moe_sk_view v; uint8_t *p = (uint8_t *)base; size_t off = 0;
    const int64_t n_pairs = n_tok * n_used;
    v.n_rows = n_tok * n_b;
    v.ntile_cap = (n_expert < n_pairs ? n_expert : n_pairs) + n_pairs / 128 + 1;          // >= sum_e ceil(cnt_e / 128)
    v.tile_rec = (int32_t *)(p + off); off += align256((size_t)v.ntile_cap * CDNA4_SK_REC * 4);
    v.wg_begin = (int32_t *)(p + off); off += align256((size_t)(1024 + 2) * 4);
    v.xh = (void *)(p + off); off += align256((size_t)v.n_rows * K * 2) + 32768;
    v.total = off;
    return v;
}
// does ggml_cdna4_mul_mat_id take the stream-k form for this call?  Q4_K experts (Q4_0 with a resident image asks with CDNA4_Q4_0R); CDNA4_MOE_SK=0: the per-tile launches
static bool moe_sk_on(int type, int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok) {
    static const bool off = getenv("CDNA4_MOE_SK") && atoi(getenv("CDNA4_MOE_SK")) == 0;
    if (off || n_expert > 1024) return false;
    const moe_sk_view v = moe_sk_carve(K, n_expert, n_used, n_b, n_tok, nullptr);
    return cdna4_gemm_sk_supported(type, M, K, v.n_rows, v.ntile_cap);
}

extern "C" {

int ggml_cdna4_api_version(void) { return GGML_CDNA4_API_VERSION; }
void ggml_cdna4_debug_trace(void *device_buffer) { cdna4_debug_trace = device_buffer; }
uint64_t ggml_cdna4_scratch_generation(void) { return cdna4_scratch_generation(); }
const char *ggml_cdna4_last_error(void) { return g_err; }
int ggml_cdna4_set_shared_device(int shared) { return cdna4_gemm_set_shared_device(shared); }
// a launch on a route that WAITS for co-resident work-groups (only chosen on an owned device) ran into its bound: its output tile(s) are NaN.  Never silent: the code the
// kernels stored (1 grid barrier, 2 / 4 hand-off, 3 reduce-scatter) comes back here, and the library has switched itself to the shared mode.
int ggml_cdna4_device_fault(int clear) { return cdna4_gemm_take_fault(clear != 0); }
// test hook: "another tenant" — work-groups that hold LDS (so that none of our 130-KB work-groups fits beside them) until the host releases them or the time is up
#ifndef CDNA4_HW_OVERRIDE
__global__ __launch_bounds__(64) void k_debug_occupy(const int *release, unsigned long long max_ticks, int lds_bytes) {
    extern __shared__ uint8_t occ_lds[];
    if (threadIdx.x == 0) occ_lds[lds_bytes - 1] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < max_ticks) {
        if (release && __hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        __builtin_amdgcn_s_sleep(32);
    }
}
#endif
int ggml_cdna4_debug_occupy(int n_workgroups, int lds_kb, const int *release, int max_ms, void *stream) {
    if (n_workgroups <= 0 || lds_kb < 1 || lds_kb > 160 || max_ms < 0 || max_ms > 20000) return cdna4_set_error_msg("debug_occupy: 1 .. 160 KB of LDS, at most 20 s");
#ifndef CDNA4_HW_OVERRIDE
    if (lds_kb > 64 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_debug_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024) != hipSuccess) { (void)hipGetLastError(); return cdna4_set_error_msg("debug_occupy: cannot raise the LDS limit"); }
    hipLaunchKernelGGL(k_debug_occupy, dim3(n_workgroups), dim3(64), lds_kb * 1024, (hipStream_t)stream, release, (unsigned long long)max_ms * 100000ull, lds_kb * 1024);
    CDNA4_CHECK_LAUNCH();
#endif
    return 0;
}
static int fault_status() {
    const int f = cdna4_gemm_take_fault(true);
    if (!f) return 0;
    snprintf(g_err, sizeof g_err, "an EARLIER launch waited for a co-resident work-group that never arrived (code %d): the device is not exclusively ours, that launch's output holds NaN tiles; "
                                  "the library now uses the non-waiting routes (ggml_cdna4_set_shared_device(1))", f);
    return -3;
}
int ggml_cdna4_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }
int ggml_cdna4_set_device(int device) { hipError_t e = hipSetDevice(device); return e == hipSuccess ? 0 : cdna4_set_error(e, __FILE__, __LINE__); }

size_t ggml_cdna4_row_size(int type, int64_t k) {
    switch (type) {
        case CDNA4_F32: return (size_t)k * 4; case CDNA4_F16: return (size_t)k * 2;
        case CDNA4_Q4_0: return k % 32 ? 0 : (size_t)(k / 32) * 18; case CDNA4_Q8_0: return k % 32 ? 0 : (size_t)(k / 32) * 34;
        case CDNA4_Q4_K: return k % 256 ? 0 : (size_t)(k / 256) * 144; case CDNA4_Q5_K: return k % 256 ? 0 : (size_t)(k / 256) * 176;
        case CDNA4_Q6_K: return k % 256 ? 0 : (size_t)(k / 256) * 210;
        case CDNA4_Q5_0: return k % 32 ? 0 : (size_t)(k / 32) * 22; case CDNA4_Q2_K: return k % 256 ? 0 : (size_t)(k / 256) * 84;
        case CDNA4_Q3_K: return k % 256 ? 0 : (size_t)(k / 256) * 110;
        case CDNA4_Q4_1: return k % 32 ? 0 : (size_t)(k / 32) * 20; case CDNA4_Q5_1: return k % 32 ? 0 : (size_t)(k / 32) * 24;
        case CDNA4_IQ4_NL: return k % 32 ? 0 : (size_t)(k / 32) * 18; case CDNA4_IQ4_XS: return k % 256 ? 0 : (size_t)(k / 256) * 136;
    }
    return 0;
}

size_t ggml_cdna4_mul_mat_workspace_size(int type, int64_t K, int64_t n_act_rows) {
    if (!is_q(type) || K <= 0 || n_act_rows <= 0) return 0;
    return carve(type, K, n_act_rows, nullptr).total;
}

size_t ggml_cdna4_mul_mat_id_workspace_size(int type, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok) {
    if (!is_q(type) || K <= 0 || n_tok <= 0 || n_used <= 0 || n_b <= 0 || n_expert <= 0) return 0;
    const size_t plain = carve(type, K, n_tok * n_b, nullptr).total;
    if (!cdna4_gemm_ids_supported(type, K) || n_tok * n_used <= 32) return plain;
    size_t grouped = moe_carve(K, n_expert, n_used, n_tok, nullptr).total;
    // (the stream-k form's carve — Q4_K, and Q4_0 in case its stack has a resident image; M is not known here: its tables do not depend on it)
    if ((type == CDNA4_Q4_K || type == CDNA4_Q4_0) && K % 256 == 0 && n_expert <= 1024) { const size_t sk = moe_sk_carve(K, n_expert, n_used, n_b, n_tok, nullptr).total; if (sk > grouped) grouped = sk; }
    return grouped > plain ? grouped : plain;
}

// the public re-encodings keep the shape (Q2_K's two-part form needs the doubled activation image: library-internal)
static inline int public_convert_target(int type) {
    static const bool any = getenv("CDNA4_DIAG_CONVERT_ANY") && atoi(getenv("CDNA4_DIAG_CONVERT_ANY")) != 0;      // diagnosis only (scripts/gpu_diag_iq4xs2.py): the two-part forms too
    return (any || cdna4_convert_weights_kmul(type) == 1) ? cdna4_convert_weights_target(type) : -1;
}
int ggml_cdna4_convert_weights_target(int type) { return public_convert_target(type); }
size_t ggml_cdna4_convert_weights_size(int type, int64_t M, int64_t K) {
    if (public_convert_target(type) < 0 || M <= 0 || K <= 0 || ggml_cdna4_row_size(type, K) == 0) return 0;
    return cdna4_convert_weights_bytes(type, M, K);
}
int ggml_cdna4_convert_weights(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, void *out, void *stream) {
    if (public_convert_target(type) < 0) return cdna4_set_error_msg("convert_weights: no exact target format for this type");
    if (M <= 0) return 0;
    if (K <= 0 || ggml_cdna4_row_size(type, K) == 0) return cdna4_set_error_msg("convert_weights: K is not a whole number of blocks");
    if (!W || !out || w_row_bytes < (int64_t)ggml_cdna4_row_size(type, K)) return cdna4_set_error_msg("convert_weights: bad pointers or row stride");
    return cdna4_launch_convert_weights(type, (const uint8_t *)W, w_row_bytes, M, K, (uint8_t *)out, (hipStream_t)stream);
}

// ---- resident kernel-native images (see gemm_q_mfma.hip: cdna4_resident_*)
size_t ggml_cdna4_resident_image_size(int type, int64_t M, int64_t K) {
    if (M <= 0 || K <= 0 || ggml_cdna4_row_size(type, K) == 0) return 0;
    const size_t rb = cdna4_resident_image_row_bytes(type, K);
    return rb ? (size_t)M * rb + 256 : 0;                              // + slack: the kernels read whole 16-byte pieces
}
__global__ void k_count_diff16(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, int64_t n16, unsigned *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const u32x4 x = a[i], y = b[i];
    if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) atomicAdd(out, 1u);
}
int ggml_cdna4_resident_image_register(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, void *image, int verify, void *stream) {
    const size_t need = ggml_cdna4_resident_image_size(type, M, K);
    if (need == 0) return cdna4_set_error_msg("resident_image: this type has no kernel-native image");
    if (!W || !image || w_row_bytes < (int64_t)ggml_cdna4_row_size(type, K) || ((uintptr_t)image & 255)) return cdna4_set_error_msg("resident_image: bad pointers, row stride or image alignment (256 bytes)");
    hipStream_t st = (hipStream_t)stream;
    int rc = cdna4_resident_build(type, (const uint8_t *)W, w_row_bytes, M, K, (uint8_t *)image, st);
    if (rc) return rc;
    if (verify) {
        // the re-encoding a second time, into library scratch, compared byte for byte: a conversion that is not bit-stable on this device (round 3 saw ~1 % of an IQ4_XS
        // product differ between identical calls; its cause was traced to the re-encoding kernel) is refused HERE, at load time
        const size_t bytes = (size_t)M * cdna4_resident_image_row_bytes(type, K), n16 = bytes / 16;
        uint8_t *again = (uint8_t *)cdna4_gemm_scratch(bytes + 512, 3);
        if (!again) return cdna4_set_error_msg("resident_image: cannot allocate the verification scratch");
        unsigned *cnt = (unsigned *)(again + ((bytes + 255) & ~(size_t)255));
        hipError_t e = hipMemsetAsync(cnt, 0, 4, st);
        if (e != hipSuccess) return cdna4_set_error(e, __FILE__, __LINE__);
        rc = cdna4_resident_build(type, (const uint8_t *)W, w_row_bytes, M, K, again, st);
        if (rc) return rc;
        if (n16) hipLaunchKernelGGL(k_count_diff16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (const u32x4 *)image, (const u32x4 *)again, (int64_t)n16, cnt);
        CDNA4_CHECK_LAUNCH();
        unsigned bad = 0;
        e = hipMemcpyAsync(&bad, cnt, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return cdna4_set_error(e, __FILE__, __LINE__);
        if (bad) { snprintf(g_err, sizeof g_err, "resident_image: two re-encodings of the same matrix differ in %u 16-byte pieces (type %d, %lld x %lld): image not registered", bad, type, (long long)M, (long long)K); return -1; }
    } else {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return cdna4_set_error(e, __FILE__, __LINE__);
    }
    return cdna4_resident_register(type, W, w_row_bytes, M, K, image);
}
int ggml_cdna4_resident_image_unregister(const void *W) { return cdna4_resident_unregister(W); }
int ggml_cdna4_resident_image_lookup(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, const void **image) {
    const uint8_t *r = cdna4_resident_lookup(type, W, w_row_bytes, M, K);
    if (image) *image = r;
    return r ? 1 : 0;
}

int ggml_cdna4_quantize_q8_K(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d, int16_t *bsums, void *xh, void *stream) {
    if (((uintptr_t)x | (uintptr_t)(x_row_stride * 4)) & 15) return cdna4_set_error_msg("quantize_q8_K: x must be 16-byte aligned");
    if (qs && (!d || !bsums)) return cdna4_set_error_msg("quantize_q8_K: qs needs d and bsums");
    return cdna4_launch_quantize_q8_K(x, x_row_stride, K, B, qs, d, bsums, xh, (hipStream_t)stream);
}
int ggml_cdna4_quantize_q8_0(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d, void *xh, int ref_rounding, void *stream) {
    if (((uintptr_t)x | (uintptr_t)(x_row_stride * 4)) & 15) return cdna4_set_error_msg("quantize_q8_0: x must be 16-byte aligned");
    if (qs && !d) return cdna4_set_error_msg("quantize_q8_0: qs needs d");
    return cdna4_launch_quantize_q8_0(x, x_row_stride, K, B, qs, d, xh, ref_rounding != 0, (hipStream_t)stream);
}

int ggml_cdna4_quantize_q8_1(const float *x, int64_t x_row_stride, int64_t K, int64_t B, int8_t *qs, float *d, float *s, void *xh, void *stream) {
    if (((uintptr_t)x | (uintptr_t)(x_row_stride * 4)) & 15) return cdna4_set_error_msg("quantize_q8_1: x must be 16-byte aligned");
    return cdna4_launch_quantize_q8_1(x, x_row_stride, K, B, qs, d, s, xh, false, (hipStream_t)stream);
}

// AUTO: one row -> the one-launch decode GEMV; 2 .. 64 rows of a format the int8 matrix-core kernel takes (mmq_i8.hip) -> that (an integer-dot
// path like the GEMV: PATH_GEMV family, workspace = the int8 SoA); above 8 rows otherwise, and above 64 always -> the fp16 MFMA GEMM
static bool use_mmq(int type, int64_t M, int64_t K, int64_t B) {
    static const bool off = getenv("CDNA4_NO_MMQ") && atoi(getenv("CDNA4_NO_MMQ")) != 0;
    static const int minb = getenv("CDNA4_MMQ_MINB") ? atoi(getenv("CDNA4_MMQ_MINB")) : 0, maxb = getenv("CDNA4_MMQ_MAXB") ? atoi(getenv("CDNA4_MMQ_MAXB")) : 0;   // measurement knobs
    if (off || !cdna4_mmq_supported(type, M, K, B)) return false;
    if (maxb > 0) return B >= (minb > 0 ? minb : 2) && B <= maxb;
    // Where it wins on MI355X — round 4: DEVICE time (HIP-graph replay of 40 calls, quantizer launch included; scripts/gpu_batch_sweep.py ->
    // profiles/r04/batch_sweep.txt: five formats x {4096^2, 4096 x 14336, 3072 x 768} x 2 .. 64 rows x {AUTO, kernel off, kernel forced}; round 3's sweep was
    // taken through a Python loop whose host time hid the differences below ~14 us).  us per call, int8 matrix cores vs the alternative:
    //   2 rows      GEMV everywhere (Q4_K 4096^2 4.8 vs 7.6)
    //   3 .. 4      GEMV unless the matrix is large: 4096^2 6.8 / 7.1 vs 7.8 / 7.9, 4096 x 14336 17.0 / 17.4 (GEMV) vs 13.8 / 14.0
    //   5 .. 8      the matrix cores EVERYWHERE: the 8-column GEMV forms take 11.2 / 11.5 us at 4096^2 (3072 x 768: 7.7 / 7.9) against 7.9 (5.9)      [new in round 4]
    //   9 .. 32     the matrix cores: 7.9 .. 11.0 vs 19.7 .. 20.6 on the fp16 GEMM at 4096^2; Q6_K 4096 x 14336 20.5 .. 27.9 vs 79
    //   33 .. 48    the matrix cores (not Q6_K: its three-group form spills): 4096^2 14.0 vs 21.1; 4096 x 14336 Q4_K 29.4 vs 31.4, Q4_0 40.1 vs 50.3, Q8_0 48.0 vs
    //               76.8, Q5_K 29.8 vs 63.6                                                                                                        [new in round 4]
    //   49 .. 64    small matrices (4096^2: 20.0 vs 21.0; 3072 x 768: 10.1 vs 13.7), and Q8_0 / Q5_K at any size (4096 x 14336: 59.9 vs 77.3, 48.3 vs 64.3); Q4_K / Q4_0
    //               on large matrices stay on the fp16 GEMM (49.0 vs 31.7, 52.8 vs 50.9)
    if (B >= 5 && B <= 48) return true;
    if (B >= 3 && B <= 4) return M * K >= ((int64_t)1 << 25);
    if (B >= 49) return M * K <= ((int64_t)1 << 24) || type == CDNA4_Q8_0 || type == CDNA4_Q5_K;
    return false;
}
static int resolve_path(int type, int path, int64_t M, int64_t K, int64_t B) {
    if (path == GGML_CDNA4_PATH_AUTO) {
        if (use_mmq(type, M, K, B)) return GGML_CDNA4_PATH_GEMV;
        return (B > 8 && cdna4_gemm_q_supported(type, M, K, B)) ? GGML_CDNA4_PATH_GEMM : GGML_CDNA4_PATH_GEMV;
    }
    return path;
}

int ggml_cdna4_prepare_act(int type, const float *X, int64_t x_row_stride, int64_t K, int64_t B, void *workspace, size_t workspace_bytes, int path, void *stream) {
    if (!is_q(type)) return cdna4_set_error_msg("prepare_act: unsupported weight type");
    if (B <= 0 || K <= 0) return 0;
    if (!workspace || ((uintptr_t)workspace & 255)) return cdna4_set_error_msg("prepare_act: workspace must be 256-byte aligned");
    const ws_view v = carve(type, K, B, workspace);
    if (workspace_bytes < v.total) return cdna4_set_error_msg("prepare_act: workspace too small");
    const bool want_i8 = path != GGML_CDNA4_PATH_GEMM, want_h = path != GGML_CDNA4_PATH_GEMV;
    if (is_kq(type)) {
        const int rc = ggml_cdna4_quantize_q8_K(X, x_row_stride, K, B, want_i8 ? v.qs : nullptr, v.d, v.bsums, want_h ? v.xh : nullptr, stream);
        if (rc || !want_h || cdna4_convert_weights_kmul(type) == 1) return rc;
        // the image is k-panel-major ([K / 128][B][128] fp16): k-panels K / 128 .. 2 K / 128 - 1 of the doubled image are a copy of the first B K halves
        const hipError_t e = hipMemcpyAsync((char *)v.xh + (size_t)B * K * 2, v.xh, (size_t)B * K * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        return e == hipSuccess ? 0 : cdna4_set_error(e, __FILE__, __LINE__);
    }
    if (!cdna4_is_q81(type)) return ggml_cdna4_quantize_q8_0(X, x_row_stride, K, B, want_i8 ? v.qs : nullptr, v.d, want_h ? v.xh : nullptr, 0, stream);
    // Q4_1 / Q5_1: Q8_1 activations — s = fp16(d * sum q) per 32-block as fp32 where the K-quants keep their bsums (same byte count); the GEMM reads the
    // fp16 image twice in a row ([d q | m 1] weights): whole 128-k panels only (cdna4_gemm_q_supported)
    if (((uintptr_t)X | (uintptr_t)(x_row_stride * 4)) & 15) return cdna4_set_error_msg("quantize_q8_1: x must be 16-byte aligned");
    // GEMM image: [x~ | s e0] — the quantizer writes both halves (the second: the CPU's own fp16 s per 32-block against the weights' [d q | m e0]);
    // a ragged last panel has no GEMM form (cdna4_gemm_q_supported says so): then only the first half is written
    return cdna4_launch_quantize_q8_1(X, x_row_stride, K, B, want_i8 ? v.qs : nullptr, v.d, reinterpret_cast<float *>(v.bsums), want_h ? v.xh : nullptr,
                                      want_h && K % 128 == 0, (hipStream_t)stream);
}

static cdna4_gemm_args gemm_args_of(int type, const void *W, int64_t w_row_bytes, const void *xh, float *Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B,
                                    int gemm_variant, int splitk, const cdna4_epilogue &epi) {
    cdna4_gemm_args a{};
    a.type = type; a.W = (const uint8_t *)W; a.w_row_bytes = w_row_bytes; a.xh = xh; a.xh_row_elems = K;
    a.Y = Y; a.y_row_elems = y_row_stride; a.M = (int)M; a.K = (int)K; a.B = (int)B; a.variant = gemm_variant; a.splitk = splitk; a.epi = epi;
    return a;
}
// epi: the MUL_MAT's tail; *tail_done says whether the launch applied it (GEMM route on k_gemm_kq_t64) — otherwise the caller appends k_epilogue
static int mul_mat_prepared_impl(int type, const void *W, int64_t w_row_bytes, float *Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B,
                                 const void *workspace, size_t workspace_bytes, int path, int gemm_variant, int splitk, const cdna4_epilogue &epi, bool *tail_done, void *stream);
int ggml_cdna4_mul_mat_prepared(int type, const void *W, int64_t w_row_bytes, float *Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B,
                                const void *workspace, size_t workspace_bytes, int path, int gemm_variant, int splitk, void *stream) {
    if (const int frc = fault_status()) return frc;
    bool done = false;
    return mul_mat_prepared_impl(type, W, w_row_bytes, Y, y_row_stride, M, K, B, workspace, workspace_bytes, path, gemm_variant, splitk, cdna4_epilogue{}, &done, stream);
}
static int mul_mat_prepared_impl(int type, const void *W, int64_t w_row_bytes, float *Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B,
                                 const void *workspace, size_t workspace_bytes, int path, int gemm_variant, int splitk, const cdna4_epilogue &epi, bool *tail_done, void *stream) {
    *tail_done = false;
    if (!is_q(type)) return cdna4_set_error_msg("mul_mat: unsupported weight type");
    if (M <= 0 || B <= 0) return 0;
    if (K <= 0 || ggml_cdna4_row_size(type, K) == 0) return cdna4_set_error_msg("mul_mat: K is not a whole number of blocks");
    const ws_view v = carve(type, K, B, (void *)workspace);
    if (workspace_bytes < v.total) return cdna4_set_error_msg("mul_mat: workspace too small");
    const bool mmq = path == GGML_CDNA4_PATH_AUTO && use_mmq(type, M, K, B) && !(((uintptr_t)W | (uintptr_t)w_row_bytes) & (type == CDNA4_Q6_K ? 1 : 15));
    path = resolve_path(type, path, M, K, B);
    if (path == GGML_CDNA4_PATH_GEMM) {
        cdna4_gemm_args a = gemm_args_of(type, W, w_row_bytes, v.xh, Y, y_row_stride, M, K, B, gemm_variant, splitk, epi);
        *tail_done = cdna4_gemm_q_fuses_tail(a);                     // (the routing itself, probed: gemm_q_mfma.hip)
        if (!*tail_done) a.epi = cdna4_epilogue{};                      // this route stores the plain product: the caller appends k_epilogue
        return cdna4_launch_gemm_q(a, (hipStream_t)stream);
    }
    cdna4_gemv_args g{};
    g.type = type; g.W = (const uint8_t *)W; g.w_row_bytes = w_row_bytes; g.qs = v.qs; g.d = v.d; g.bsums = v.bsums;
    g.Y = Y; g.y_col_stride = y_row_stride; g.M = (int)M; g.K = (int)K; g.ncol = (int)B; g.ids = nullptr; g.epi = epi;
    *tail_done = true;
    if (mmq) return cdna4_launch_mmq(g, (hipStream_t)stream);
    return cdna4_launch_gemv_q(g, (hipStream_t)stream);
}

static int mul_mat_impl(int type, const void *W, int64_t w_row_bytes, const float *X, int64_t x_row_stride, float *Y, int64_t y_row_stride,
                        int64_t M, int64_t K, int64_t B, void *workspace, size_t workspace_bytes, int path, int gemm_variant, int splitk, const cdna4_epilogue &epi, void *stream) {
    if (!is_q(type)) return cdna4_set_error_msg("mul_mat: unsupported weight type");
    if (M <= 0 || B <= 0) return 0;
    if (K <= 0 || ggml_cdna4_row_size(type, K) == 0) return cdna4_set_error_msg("mul_mat: K is not a whole number of blocks");
    // Formats that reach the MFMA kernels through an exact re-encoding (Q5_0 / IQ4_NL -> Q8_0, convert_w.hip) follow their TARGET format onto the int8
    // matrix-core kernel at the batch sizes where that one is chosen: re-encode into scratch, then the target's route (its integer dots on Q8_0
    // activations are the CPU's own arithmetic for the source format: vec_dot_q5_0_q8_0 / vec_dot_iq4_nl_q8_0 on the same quants).  Same result as
    // ggml_cdna4_convert_weights up front + ggml_cdna4_mul_mat on the target format (tests/test_gpu_widening.py).
    if (path == GGML_CDNA4_PATH_AUTO && cdna4_convert_weights_kmul(type) == 1) {
        const int tgt = cdna4_convert_weights_target(type);
        // (from 9 rows up: below that the re-encoding pass — ~2.5x the weight bytes — costs more than the staged GEMV it would replace)
        if (tgt >= 0 && tgt != type && B >= 9 && use_mmq(tgt, M, K, B)) {
            uint8_t *cw = const_cast<uint8_t *>(cdna4_resident_lookup(type, W, w_row_bytes, M, K));      // a resident image (built once at load): no conversion launch
            if (!cw) {
                cw = (uint8_t *)cdna4_gemm_scratch(cdna4_convert_weights_bytes(type, M, K) + 256, 3);
                if (!cw) return cdna4_set_error_msg("mul_mat: cannot allocate the re-encoded weights");
                const int rc = cdna4_launch_convert_weights(type, (const uint8_t *)W, w_row_bytes, M, K, cw, (hipStream_t)stream);
                if (rc) return rc;
            }
            return mul_mat_impl(tgt, cw, (int64_t)ggml_cdna4_row_size(tgt, K), X, x_row_stride, Y, y_row_stride, M, K, B, workspace, workspace_bytes, path, gemm_variant, splitk, epi, stream);
        }
    }
    // the int8 matrix-core kernel: AUTO only (an explicit PATH_GEMV keeps the v_dot4 units — tests compare the two), aligned Q4_K rows
    const bool mmq = path == GGML_CDNA4_PATH_AUTO && use_mmq(type, M, K, B) && !(((uintptr_t)W | (uintptr_t)w_row_bytes) & (type == CDNA4_Q6_K ? 1 : 15));
    path = resolve_path(type, path, M, K, B);
    if (path == GGML_CDNA4_PATH_GEMM && !cdna4_gemm_q_supported(type, M, K, B)) return cdna4_set_error_msg("mul_mat: GEMM path does not support this shape");
    // one launch while the redundant per-work-group quantization is cheap (B x K up to 32 K values, rounded to the instantiated 2 / 4 / 8 columns)
    const int64_t nbt = B <= 1 ? 1 : (B <= 2 ? 2 : (B <= 4 ? 4 : 8));
    if (!mmq && path == GGML_CDNA4_PATH_GEMV && cdna4_gemv_fused_supported(type, K, B) && (B == 1 || nbt * K <= 32768) && !(((uintptr_t)X | (uintptr_t)(B > 1 ? x_row_stride * 4 : 0)) & 15)) {
        // decode with 1..8 activation rows: the activation quantizer runs inside the GEMV kernel (workspace untouched), ONE launch
        cdna4_gemv_args g{};
        g.type = type; g.W = (const uint8_t *)W; g.w_row_bytes = w_row_bytes; g.Y = Y; g.y_col_stride = y_row_stride;
        g.M = (int)M; g.K = (int)K; g.ncol = (int)B; g.ids = nullptr; g.epi = epi;
        return B == 1 ? cdna4_launch_gemv_q_fused(g, X, (hipStream_t)stream) : cdna4_launch_gemv_q_fused_n(g, X, x_row_stride, (hipStream_t)stream);
    }
    if (path == GGML_CDNA4_PATH_GEMM && workspace && !((uintptr_t)workspace & 255)) {
        // ONE launch for the whole step where the routing says so (k_gemm_kq_t64<.., FQ>: every work-group quantizes its share of X into the image in the workspace,
        // a grid barrier, then the multiply — what ggml_compute_forward_mul_mat does inside one op, ggml-cpu.c:7490-7509 + :7428-7605); same sums, same order as the
        // two launches below.  The routing code itself answers (probed: no side effects).
        const ws_view v = carve(type, K, B, workspace);
        if (workspace_bytes < v.total) return cdna4_set_error_msg("mul_mat: workspace too small");
        cdna4_gemm_args a = gemm_args_of(type, W, w_row_bytes, v.xh, Y, y_row_stride, M, K, B, gemm_variant, splitk, epi);
        a.xf = X; a.xf_row_elems = x_row_stride;
        if (cdna4_gemm_q_fuses_quantizer(a)) {
            const bool tail_done = cdna4_gemm_q_fuses_tail(a);
            if (!tail_done) a.epi = cdna4_epilogue{};
            const int rc1 = cdna4_launch_gemm_q(a, (hipStream_t)stream);
            if (rc1 || tail_done) return rc1;
            return cdna4_launch_epilogue(Y, y_row_stride, M, B, epi, (hipStream_t)stream);
        }
    }
    int rc = ggml_cdna4_prepare_act(type, X, x_row_stride, K, B, workspace, workspace_bytes, path, stream);
    if (rc) return rc;
    if (path == GGML_CDNA4_PATH_GEMV) {                                  // a few activation rows: the tail rides in the GEMV's store
        const ws_view v = carve(type, K, B, workspace);
        cdna4_gemv_args g{};
        g.type = type; g.W = (const uint8_t *)W; g.w_row_bytes = w_row_bytes; g.qs = v.qs; g.d = v.d; g.bsums = v.bsums;
        g.Y = Y; g.y_col_stride = y_row_stride; g.M = (int)M; g.K = (int)K; g.ncol = (int)B; g.ids = nullptr; g.epi = epi;
        if (mmq) return cdna4_launch_mmq(g, (hipStream_t)stream);                                                   // 2..64 rows: int8 MFMA
        if (cdna4_gemv_staged_supported(type, K, B)) return cdna4_launch_gemv_q_staged(g, (hipStream_t)stream);    // 2..8 rows: columns from LDS
        return cdna4_launch_gemv_q(g, (hipStream_t)stream);
    }
    bool tail_done = false;
    rc = mul_mat_prepared_impl(type, W, w_row_bytes, Y, y_row_stride, M, K, B, workspace, workspace_bytes, path, gemm_variant, splitk, epi, &tail_done, stream);
    if (rc || tail_done) return rc;                                      // Q4_K: the tail went out with k_gemm_kq_t64's store
    return cdna4_launch_epilogue(Y, y_row_stride, M, B, epi, (hipStream_t)stream);       // the older GEMM kernels: one element-wise launch for the whole tail
}
int ggml_cdna4_mul_mat(int type, const void *W, int64_t w_row_bytes, const float *X, int64_t x_row_stride, float *Y, int64_t y_row_stride,
                       int64_t M, int64_t K, int64_t B, void *workspace, size_t workspace_bytes, int path, int gemm_variant, int splitk, void *stream) {
    if (const int frc = fault_status()) return frc;
    return mul_mat_impl(type, W, w_row_bytes, X, x_row_stride, Y, y_row_stride, M, K, B, workspace, workspace_bytes, path, gemm_variant, splitk, cdna4_epilogue{}, stream);
}
// n MUL_MATs of ONE activation row (B = 1) in ONE launch: wq / wk / wv, w_gate / w_up of a decoded token (round 6).  Matrices of one type and K; bias[i] may be null.
// Bit-identical to n ggml_cdna4_mul_mat[_fused] calls with B = 1 (the same kernel body per matrix).  Returns -2 where the call has no grouped form (the caller issues the
// separate calls): types outside the five headline formats, a K the one-launch decode does not take, misaligned rows.
int ggml_cdna4_mul_mat_group(int type, int n, const void *const *W, const int64_t *w_row_bytes, const int64_t *M, float *const *Y, const float *const *bias, const float *X, int64_t K, void *stream) {
    if (const int frc = fault_status()) return frc;
    if (n < 1 || n > CDNA4_GEMV_GROUP_MAX || !W || !w_row_bytes || !M || !Y || !X) { cdna4_set_error_msg("mul_mat_group: 1 .. 4 matrices"); return -2; }
    if (!(type == CDNA4_Q4_K || type == CDNA4_Q5_K || type == CDNA4_Q6_K || type == CDNA4_Q4_0 || type == CDNA4_Q8_0) || K <= 0 || ggml_cdna4_row_size(type, K) == 0 ||
        !cdna4_gemv_fused_supported(type, K, 1) || ((uintptr_t)X & 15)) { cdna4_set_error_msg("mul_mat_group: no grouped form for this type / K / alignment"); return -2; }
    cdna4_gemv_group g{};
    g.K = (int)K;
    for (int i = 0; i < n; i++) {
        if (M[i] <= 0 || M[i] > 0x7fffffff || !W[i] || !Y[i] || (((uintptr_t)W[i] | (uintptr_t)w_row_bytes[i]) & ((type == CDNA4_Q4_K || type == CDNA4_Q5_K) ? 15 : 1))) { cdna4_set_error_msg("mul_mat_group: bad or misaligned matrix"); return -2; }
        if (use_mmq(type, M[i], K, 1) || resolve_path(type, GGML_CDNA4_PATH_AUTO, M[i], K, 1) != GGML_CDNA4_PATH_GEMV) { cdna4_set_error_msg("mul_mat_group: a matrix whose single call is not the one-launch GEMV"); return -2; }
        g.W[i] = (const uint8_t *)W[i]; g.w_row_bytes[i] = w_row_bytes[i]; g.Y[i] = Y[i]; g.M[i] = (int)M[i]; g.bias[i] = bias ? bias[i] : nullptr;
    }
    return cdna4_launch_gemv_q_fused_grp(type, g, n, X, (hipStream_t)stream);
}
// the GEMV routes (B <= 8) apply the tail where the element is reduced, and so does the Q4_K GEMM (k_gemm_kq_t64, on 16-byte-aligned rows: what every
// ggml buffer of the plug-in and every torch allocation gives); the older MFMA GEMM kernels write the product first (k_epilogue behind them)
int ggml_cdna4_mul_mat_fused_residual_may_alias(int type, int64_t M, int64_t K, int64_t B) {
    if (resolve_path(type, GGML_CDNA4_PATH_AUTO, M, K, B) == GGML_CDNA4_PATH_GEMV) return 1;
    // GEMM route: yes where the kernel AUTO picks carries the tail in its store (every element is read and written by the same lane) — asked of the routing
    // itself, for contiguous 256-byte-aligned operands (the call re-checks with the real pointers)
    cdna4_epilogue e{}; e.act = 1;
    return cdna4_gemm_q_fuses_tail(gemm_args_of(type, (const void *)(uintptr_t)256, (int64_t)ggml_cdna4_row_size(type, K), (const void *)(uintptr_t)256, (float *)(uintptr_t)256, M, M, K, B, 0, 0, e)) ? 1 : 0;
}
// the route ggml_cdna4_mul_mat(path = AUTO) takes for a contiguous, 256-byte-aligned call of this shape on the current device — host logic only, no launch, no scratch:
//   1 one launch (activation quantizer inside the GEMV)     2 quantize + GEMV (columns staged in LDS)     3 quantize + int8 matrix-core kernel
//   10 quantize + k_gemm_kq_t64     11 ONE launch: the quantizer inside k_gemm_kq_t64 (resident grids on an owned device)     12 + k_gemm_r8     13 + a 128 x 128-tile kernel     14 + an older per-lane-load GEMM;   + 100: behind an exact re-encoding of the weights
static int route_of(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, int64_t B);
int ggml_cdna4_mul_mat_route(int type, int64_t M, int64_t K, int64_t B) {
    return route_of(type, (const void *)(uintptr_t)256, (int64_t)ggml_cdna4_row_size(type, K), M, K, B);
}
// the same for a concrete weight matrix: sees its alignment, its row stride and a RESIDENT image registered for it (Q4_0 with an image: 10 / 12 instead of 13)
int ggml_cdna4_mul_mat_route_of(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, int64_t B) {
    if (!W || w_row_bytes < (int64_t)ggml_cdna4_row_size(type, K)) return 0;
    return route_of(type, W, w_row_bytes, M, K, B);
}
static int route_of(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, int64_t B) {
    if (!is_q(type) || M <= 0 || B <= 0 || K <= 0 || ggml_cdna4_row_size(type, K) == 0) return 0;
    if (cdna4_convert_weights_kmul(type) == 1) {
        const int tgt = cdna4_convert_weights_target(type);
        if (tgt >= 0 && tgt != type && B >= 9 && use_mmq(tgt, M, K, B)) return 103;
    }
    if (use_mmq(type, M, K, B)) return 3;
    const int path = resolve_path(type, GGML_CDNA4_PATH_AUTO, M, K, B);
    if (path == GGML_CDNA4_PATH_GEMV) {
        const int64_t nbt = B <= 1 ? 1 : (B <= 2 ? 2 : (B <= 4 ? 4 : 8));
        return (cdna4_gemv_fused_supported(type, K, B) && (B == 1 || nbt * K <= 32768)) ? 1 : 2;
    }
    cdna4_gemm_args ra = gemm_args_of(type, W, w_row_bytes, (const void *)(uintptr_t)256, (float *)(uintptr_t)256, M, M, K, B, 0, 0, cdna4_epilogue{});
    ra.xf = (const float *)(uintptr_t)256; ra.xf_row_elems = K;       // (the call hands its fp32 rows over: the one-launch step is a candidate)
    return cdna4_gemm_q_route(ra);
}
// the checks ggml_cdna4_mul_mat_fused and its pre-quantized twin share: 0, or the error status
static int fused_tail_ok(int type, const void *W, int64_t w_row_bytes, float *Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B, int act,
                         const float *residual, int64_t residual_row_stride, const void *workspace) {
    if (act != 0 && act != 1) return cdna4_set_error_msg("mul_mat_fused: act is 0 (none) or 1 (GELU)");
    if (residual && M > 0 && B > 0) {
        const char *r0 = (const char *)residual, *r1 = (const char *)(residual + (B - 1) * residual_row_stride + M);
        const char *y0 = (const char *)Y, *y1 = (const char *)(Y + (B - 1) * y_row_stride + M);
        const bool overlap = r0 < y1 && y0 < r1, exact = residual == Y && residual_row_stride == y_row_stride;
        bool ok = exact && ggml_cdna4_mul_mat_fused_residual_may_alias(type, M, K, B);
        if (ok && resolve_path(type, GGML_CDNA4_PATH_AUTO, M, K, B) == GGML_CDNA4_PATH_GEMM)     // the query cannot see the pointers: the fused GEMM needs aligned rows
            ok = cdna4_gemm_q_fuses_tail(gemm_args_of(type, W, w_row_bytes, workspace, Y, y_row_stride, M, K, B, 0, 0, cdna4_epilogue{}));
        if (overlap && !ok)
            return cdna4_set_error_msg("mul_mat_fused: residual overlaps Y (only an exact alias is allowed, and only where ggml_cdna4_mul_mat_fused_residual_may_alias says so)");
    }
    return 0;
}
int ggml_cdna4_mul_mat_fused(int type, const void *W, int64_t w_row_bytes, const float *X, int64_t x_row_stride, float *Y, int64_t y_row_stride,
                             int64_t M, int64_t K, int64_t B, const float *bias, int act, const float *residual, int64_t residual_row_stride,
                             void *workspace, size_t workspace_bytes, void *stream) {
    if (const int frc = fault_status()) return frc;
    if (const int rc = fused_tail_ok(type, W, w_row_bytes, Y, y_row_stride, M, K, B, act, residual, residual_row_stride, workspace)) return rc;
    cdna4_epilogue e{}; e.bias = bias; e.resid = residual; e.resid_row_stride = residual_row_stride; e.act = act;
    return mul_mat_impl(type, W, w_row_bytes, X, x_row_stride, Y, y_row_stride, M, K, B, workspace, workspace_bytes, GGML_CDNA4_PATH_AUTO, 0, 0, e, stream);
}
// ---- the graph-level hand-off of quantized activations (round 5).  A transformer layer multiplies the SAME activations by several weight matrices
// (wq / wk / wv; w_gate / w_up): the CPU backend quantizes src1 once per MUL_MAT node (ggml-cpu.c:7490-7509) and so did every ggml_cdna4_mul_mat call.  A host that knows
// the activations have not changed since the previous call on this workspace asks for the key of the image each call leaves there; equal non-zero keys (and equal X, row
// stride, K, B) mean the second call can be ggml_cdna4_mul_mat_prepared[_fused] with path AUTO — the same kernels on the same image, bit-identical, one launch fewer.
//   0: the call leaves no reusable image (one-launch decode / few-row forms quantize inside the kernel; re-encoded formats on the int8 matrix cores use their target's image)
uint32_t ggml_cdna4_act_image_key(int type, int64_t M, int64_t K, int64_t B) {
    if (!is_q(type) || M <= 0 || B <= 0 || K <= 0 || ggml_cdna4_row_size(type, K) == 0) return 0;
    if (cdna4_convert_weights_kmul(type) == 1) {
        const int tgt = cdna4_convert_weights_target(type);
        if (tgt >= 0 && tgt != type && B >= 9 && use_mmq(tgt, M, K, B)) return 0;
    }
    const bool mmq = use_mmq(type, M, K, B);
    const int path = resolve_path(type, GGML_CDNA4_PATH_AUTO, M, K, B);
    if (path != GGML_CDNA4_PATH_GEMM && !mmq) return 0;                 // (the GEMV forms below the matrix-core kernels: one launch, or their own staged variant)
    return 1u | (is_kq(type) ? 2u : 0u) | (cdna4_is_q81(type) ? 4u : 0u) | (cdna4_convert_weights_kmul(type) == 2 ? 8u : 0u) | (path == GGML_CDNA4_PATH_GEMM ? 16u : 0u);
}
// the key of the call on a CONCRETE weight matrix: a few-row call (int8 class) on rows that are not 16-byte aligned does not take the int8 matrix-core kernel — it may
// quantize inside a one-launch GEMV and leave the workspace untouched (mul_mat_impl) — so such a call names no image
uint32_t ggml_cdna4_act_image_key_of(int type, const void *W, int64_t w_row_bytes, int64_t M, int64_t K, int64_t B) {
    const uint32_t key = ggml_cdna4_act_image_key(type, M, K, B);
    if (key && !(key & 16u) && (((uintptr_t)W | (uintptr_t)w_row_bytes) & (type == CDNA4_Q6_K ? 1 : 15))) return 0;
    return key;
}
// NORM / RMS_NORM [* gain] [+ shift] that ALSO leaves, in `workspace`, the activation image a following ggml_cdna4_mul_mat(type, .., X = dst, K = dst->ne[0], B = its rows)
// would build — for the calls whose ggml_cdna4_act_image_key is the K-quants' fp16 GEMM image (19): the MUL_MATs of dst then run ggml_cdna4_mul_mat_prepared[_fused] and
// the graph pays no quantizer launch for them at all.  dst's fp32 rows are written exactly as ggml_cdna4_op_norm_affine writes them; the image is bit-identical to
// ggml_cdna4_prepare_act's.  Rows of 256 .. 8192 values, dst rows contiguous ([B][K]).
static int norm_affine_act(const ggml_cdna4_tensor *src0, const ggml_cdna4_tensor *gain, const ggml_cdna4_tensor *shift, const ggml_cdna4_tensor *dst, float eps, int rms,
                           int type, void *workspace, size_t workspace_bytes, void *stream, bool kq);
int ggml_cdna4_op_norm_affine_q8_K(const ggml_cdna4_tensor *src0, const ggml_cdna4_tensor *gain, const ggml_cdna4_tensor *shift, const ggml_cdna4_tensor *dst, float eps, int rms,
                                   int type, void *workspace, size_t workspace_bytes, void *stream) {
    if (!is_q(type) || !is_kq(type) || cdna4_convert_weights_kmul(type) != 1) return cdna4_set_error_msg("norm_q8_K: the image is the K-quants' (Q8_K activations, single image)");
    return norm_affine_act(src0, gain, shift, dst, eps, rms, type, workspace, workspace_bytes, stream, true);
}
// the same for the 32-block formats whose activations are Q8_0 (Q4_0 / Q8_0 / Q5_0 / IQ4_NL; ggml_cdna4_act_image_key == 17): the image of k_quantize_q8_0, bit for bit
int ggml_cdna4_op_norm_affine_q8_0(const ggml_cdna4_tensor *src0, const ggml_cdna4_tensor *gain, const ggml_cdna4_tensor *shift, const ggml_cdna4_tensor *dst, float eps, int rms,
                                   int type, void *workspace, size_t workspace_bytes, void *stream) {
    if (!is_q(type) || is_kq(type) || cdna4_is_q81(type) || cdna4_convert_weights_kmul(type) != 1) return cdna4_set_error_msg("norm_q8_0: the image is the 32-block formats' (Q8_0 activations, single image)");
    return norm_affine_act(src0, gain, shift, dst, eps, rms, type, workspace, workspace_bytes, stream, false);
}
static int norm_affine_act(const ggml_cdna4_tensor *src0, const ggml_cdna4_tensor *gain, const ggml_cdna4_tensor *shift, const ggml_cdna4_tensor *dst, float eps, int rms,
                           int type, void *workspace, size_t workspace_bytes, void *stream, bool kq) {
    const int64_t K = dst->ne[0], B = dst->ne[1] * dst->ne[2] * dst->ne[3];
    if (K <= 0 || B <= 0 || K % (kq ? 256 : 32)) return cdna4_set_error_msg("norm + activation image: rows of whole blocks");
    if (dst->nb[1] != K * 4 || (dst->ne[2] > 1 && dst->nb[2] != dst->ne[1] * dst->nb[1]) || (dst->ne[3] > 1 && dst->nb[3] != dst->ne[2] * dst->nb[2])) return cdna4_set_error_msg("norm_q8_K: dst rows must be contiguous");
    if (!workspace || ((uintptr_t)workspace & 255)) return cdna4_set_error_msg("norm_q8_K: workspace must be 256-byte aligned");
    const ws_view v = carve(type, K, B, workspace);
    if (workspace_bytes < v.total) return cdna4_set_error_msg("norm_q8_K: workspace too small");
    return cdna4_launch_norm_affine_q8_K(src0, gain, shift, dst, eps, rms, v.xh, stream, kq ? 1 : 0);
}
// ggml_cdna4_mul_mat_fused on the image the previous call left in `workspace` (ggml_cdna4_act_image_key)
int ggml_cdna4_mul_mat_prepared_fused(int type, const void *W, int64_t w_row_bytes, float *Y, int64_t y_row_stride, int64_t M, int64_t K, int64_t B,
                                      const float *bias, int act, const float *residual, int64_t residual_row_stride,
                                      const void *workspace, size_t workspace_bytes, void *stream) {
    if (const int frc = fault_status()) return frc;
    if (const int rc = fused_tail_ok(type, W, w_row_bytes, Y, y_row_stride, M, K, B, act, residual, residual_row_stride, workspace)) return rc;
    if (!ggml_cdna4_act_image_key(type, M, K, B)) return cdna4_set_error_msg("mul_mat_prepared_fused: a call of this shape has no prepared form (ggml_cdna4_act_image_key == 0)");
    cdna4_epilogue e{}; e.bias = bias; e.resid = residual; e.resid_row_stride = residual_row_stride; e.act = act;
    bool tail_done = false;
    const int rc = mul_mat_prepared_impl(type, W, w_row_bytes, Y, y_row_stride, M, K, B, workspace, workspace_bytes, GGML_CDNA4_PATH_AUTO, 0, 0, e, &tail_done, stream);
    if (rc || tail_done) return rc;
    return cdna4_launch_epilogue(Y, y_row_stride, M, B, e, (hipStream_t)stream);
}

// the work-queue form's choice for a call (shared by the call, its `prepared` twin and the key): Q4_K experts, or Q4_0 experts whose stack has a resident Q4_0R image
struct moe_sk_pick { int type = -1; const uint8_t *W = nullptr; int64_t row = 0, exp = 0; };
static moe_sk_pick moe_sk_pick_of(int type, const void *as, int64_t w_row_bytes, int64_t w_expert_bytes, int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok) {
    moe_sk_pick r; r.W = (const uint8_t *)as; r.row = w_row_bytes; r.exp = w_expert_bytes;
    if (type == CDNA4_Q4_K && !(((uintptr_t)as | (uintptr_t)w_row_bytes | (uintptr_t)w_expert_bytes) & 15)) r.type = CDNA4_Q4_K;
    else if (type == CDNA4_Q4_0 && K % 256 == 0 && w_expert_bytes == M * w_row_bytes) {
        const uint8_t *img = cdna4_resident_lookup(CDNA4_Q4_0, as, w_row_bytes, M * n_expert, K);
        if (img && !((uintptr_t)img & 15)) { r.type = CDNA4_Q4_0R; r.W = img; r.row = (K / 256) * 144; r.exp = M * r.row; }
    }
    // (CDNA4_MMQ_IDS=2 — a coverage / measurement knob — asks for the int8 matrix-core kernels where there are few rows per expert)
    static const bool mmq_ids_forced = getenv("CDNA4_MMQ_IDS") && atoi(getenv("CDNA4_MMQ_IDS")) == 2;
    if (mmq_ids_forced && n_tok * n_used <= 32 * n_expert) r.type = -1;
    if (r.type >= 0 && !moe_sk_on(r.type, M, K, n_expert, n_used, n_b, n_tok)) r.type = -1;
    return r;
}
static bool moe_grouped_call(int type, const void *as, int64_t w_row_bytes, int64_t w_expert_bytes, const float *b, int64_t b_row_stride, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_tok, const void *workspace) {
    return cdna4_gemm_ids_supported(type, K) && n_tok * n_used > 32 && n_expert <= 1024 && workspace && !((uintptr_t)workspace & 255) &&
           !(((uintptr_t)as | (uintptr_t)w_row_bytes | (uintptr_t)w_expert_bytes) & 1) && !(((uintptr_t)b | (uintptr_t)(b_row_stride * 4)) & 15);
}
static int mul_mat_id_impl(bool front_in_place, int type, const void *as, int64_t w_row_bytes, int64_t w_expert_bytes, const float *b, int64_t b_row_stride, int64_t b_tok_stride,
                           const int32_t *ids, int64_t ids_tok_stride, float *dst, int64_t dst_row_stride, int64_t dst_tok_stride,
                           int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok,
                           void *workspace, size_t workspace_bytes, void *stream);
int ggml_cdna4_mul_mat_id(int type, const void *as, int64_t w_row_bytes, int64_t w_expert_bytes, const float *b, int64_t b_row_stride, int64_t b_tok_stride,
                          const int32_t *ids, int64_t ids_tok_stride, float *dst, int64_t dst_row_stride, int64_t dst_tok_stride,
                          int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok,
                          void *workspace, size_t workspace_bytes, void *stream) {
    return mul_mat_id_impl(false, type, as, w_row_bytes, w_expert_bytes, b, b_row_stride, b_tok_stride, ids, ids_tok_stride, dst, dst_row_stride, dst_tok_stride, M, K, n_expert, n_used, n_b, n_tok,
                           workspace, workspace_bytes, stream);
}
// MUL_MAT_IDs of ONE (b, ids) — a mixture-of-experts layer's w_up and w_gate stacks (llama.cpp build_moe_ffn) — share everything the first launch of the work-queue form makes:
// the sorted ids, the tile records, the spans, the quantized activations.  Non-zero: a ggml_cdna4_mul_mat_id of this call leaves that FRONT in its workspace, and
// ggml_cdna4_mul_mat_id_prepared with other expert weights of the same type / M / K, the same b, ids, strides and counts, on the untouched workspace multiplies it again
// (ONE launch instead of two, bit-identical to the full call).
uint32_t ggml_cdna4_mul_mat_id_front_key(int type, const void *as, int64_t w_row_bytes, int64_t w_expert_bytes, int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok, size_t workspace_bytes) {
    if (!is_q(type) || M <= 0 || K <= 0 || n_tok <= 0 || n_used <= 0 || n_b <= 0 || n_used % n_b || ggml_cdna4_row_size(type, K) == 0) return 0;
    if (!(cdna4_gemm_ids_supported(type, K) && n_tok * n_used > 32 && n_expert <= 1024) || (((uintptr_t)as | (uintptr_t)w_row_bytes | (uintptr_t)w_expert_bytes) & 1)) return 0;
    const moe_sk_pick pk = moe_sk_pick_of(type, as, w_row_bytes, w_expert_bytes, M, K, n_expert, n_used, n_b, n_tok);
    if (pk.type < 0 || workspace_bytes < moe_sk_carve(K, n_expert, n_used, n_b, n_tok, nullptr).total) return 0;
    return 0x4D510000u | (pk.type == CDNA4_Q4_K ? 1u : 2u);            // (the image's activation class; the rest of what the front depends on is what the caller compares)
}
int ggml_cdna4_mul_mat_id_prepared(int type, const void *as, int64_t w_row_bytes, int64_t w_expert_bytes, const float *b, int64_t b_row_stride, int64_t b_tok_stride,
                                   const int32_t *ids, int64_t ids_tok_stride, float *dst, int64_t dst_row_stride, int64_t dst_tok_stride,
                                   int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!ggml_cdna4_mul_mat_id_front_key(type, as, w_row_bytes, w_expert_bytes, M, K, n_expert, n_used, n_b, n_tok, workspace_bytes)) { cdna4_set_error_msg("mul_mat_id_prepared: a call of this shape leaves no front (ggml_cdna4_mul_mat_id_front_key == 0)"); return -2; }
    return mul_mat_id_impl(true, type, as, w_row_bytes, w_expert_bytes, b, b_row_stride, b_tok_stride, ids, ids_tok_stride, dst, dst_row_stride, dst_tok_stride, M, K, n_expert, n_used, n_b, n_tok,
                           workspace, workspace_bytes, stream);
}
static int mul_mat_id_impl(bool front_in_place, int type, const void *as, int64_t w_row_bytes, int64_t w_expert_bytes, const float *b, int64_t b_row_stride, int64_t b_tok_stride,
                           const int32_t *ids, int64_t ids_tok_stride, float *dst, int64_t dst_row_stride, int64_t dst_tok_stride,
                           int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_b, int64_t n_tok,
                           void *workspace, size_t workspace_bytes, void *stream) {
    if (const int frc = fault_status()) return frc;
    if (!is_q(type)) return cdna4_set_error_msg("mul_mat_id: unsupported weight type");
    if (M <= 0 || n_tok <= 0 || n_used <= 0) return 0;
    if (K <= 0 || ggml_cdna4_row_size(type, K) == 0) return cdna4_set_error_msg("mul_mat_id: K is not a whole number of blocks");
    if (n_b <= 0 || n_used % n_b) return cdna4_set_error_msg("mul_mat_id: n_used must be a multiple of b.ne[1]");   // ggml_mul_mat_id: ids->ne[0] % b->ne[1] == 0 (src/ggml.c:2747); slot u reads row u % n_b
    if (dst_tok_stride != n_used * dst_row_stride) return cdna4_set_error_msg("mul_mat_id: dst must be contiguous over (slot, token)");
    if (b_tok_stride != n_b * b_row_stride) return cdna4_set_error_msg("mul_mat_id: b must be contiguous over (row, token)");
    const int64_t nact = n_tok * n_b;
    // prefill-sized mixture-of-experts batches: group the (token, slot) rows by expert on the device and run ONE MFMA GEMM launch over
    // the (expert, activation tile) table — every expert's weights are read once per m-tile instead of once per column
    // (ggml_compute_forward_mul_mat_id groups the same way on the host: ggml-cpu.c:7648-7781)
    if (front_in_place && !moe_grouped_call(type, as, w_row_bytes, w_expert_bytes, b, b_row_stride, K, n_expert, n_used, n_tok, workspace)) { cdna4_set_error_msg("mul_mat_id_prepared: misaligned operands"); return -2; }
    if (moe_grouped_call(type, as, w_row_bytes, w_expert_bytes, b, b_row_stride, K, n_expert, n_used, n_tok, workspace)) {
        // Q4_K experts, and Q4_0 experts whose stack has a RESIDENT Q4_0R image (ggml's 3-D expert tensor, rows back to back; found by the stack's pointer): the stream-k form —
        // TWO launches (planner + token-order quantizer; one persistent grouped GEMM that gathers its rows), round 6
        {
            const moe_sk_pick pk = moe_sk_pick_of(type, as, w_row_bytes, w_expert_bytes, M, K, n_expert, n_used, n_b, n_tok);
            const int sk_type = pk.type; const uint8_t *skW = pk.W; const int64_t sk_row = pk.row, sk_exp = pk.exp;
            if (sk_type >= 0) {
                const moe_sk_view sv = moe_sk_carve(K, n_expert, n_used, n_b, n_tok, workspace);
                if (workspace_bytes >= sv.total) {
                    // cost of one (tile, m-tile, superblock) unit by the tile's fragments in use (1 .. 4), relative.  FLAT, by measurement: the k loop is bound by its unpack /
                    // issue work, which does not depend on the fragments — a tile with one fragment in use takes 1.78-1.91 us per superblock, a full one 1.86-1.92
                    // (profiles/r06/moe_sk_trace_*.txt), with or without the activation DMA of the empty fragments; weights that call light tiles cheaper lost: 8 x 2 x 512 x 4096^2,
                    // us per call, one box: 10,10,10,10 72.2 | 8,9,10,10 79.8 | 7,8,9,10 85.0 | 6,8,9,10 90.7 | 5,7,9,10 98.1 | 4,6,8,10 107.6 (CDNA4_SK_CW=a,b,c,d: measurement knob)
                    struct sk_costs { int c[4]; sk_costs() : c{10, 10, 10, 10} { int t4[4]; const char *ev = getenv("CDNA4_SK_CW");
                        if (ev && sscanf(ev, "%d,%d,%d,%d", &t4[0], &t4[1], &t4[2], &t4[3]) == 4 && t4[0] > 0 && t4[1] > 0 && t4[2] > 0 && t4[3] > 0 && t4[3] < 256) for (int i = 0; i < 4; i++) c[i] = t4[i]; } };
                    static const sk_costs costs;
                    const int *cw = costs.c;
                    const int G = cdna4_gemm_sk_spans();
                    const int64_t upt = ((M + 127) / 128) * (K / 256);
                    // (prepared: the front of an earlier call with the same b, ids and shape is in place)
                    int rc = front_in_place ? 0 : cdna4_launch_moe_sk_front(ids, ids_tok_stride, (int)n_tok, (int)n_used, (int)n_b, (int)n_expert, (int)sv.ntile_cap, (int)upt, G, cw, sv.tile_rec, sv.wg_begin,
                                                                            b, b_row_stride, K, sk_type == CDNA4_Q4_K, sv.xh, (hipStream_t)stream);
                    if (rc) return rc;
                    cdna4_gemm_args a{};
                    a.type = sk_type; a.W = skW; a.w_row_bytes = sk_row; a.xh = sv.xh; a.xh_row_elems = K;
                    a.Y = dst; a.y_row_elems = dst_row_stride; a.M = (int)M; a.K = (int)K; a.B = (int)sv.n_rows;
                    return cdna4_launch_gemm_sk(a, sv.tile_rec, sv.wg_begin, G, sk_exp, (hipStream_t)stream);
                }
            }
        }
        if (front_in_place) { cdna4_set_error_msg("mul_mat_id_prepared: this call does not take the work-queue form"); return -2; }
        const moe_view mv = moe_carve(K, n_expert, n_used, n_tok, workspace);
        if (workspace_bytes >= mv.total && mv.img_rows * K * 2 < ((int64_t)1 << 31)) {
            int rc = cdna4_launch_moe_plan(ids, ids_tok_stride, (int)n_tok, (int)n_used, (int)n_b, (int)n_expert, (int)mv.img_rows, mv.img_src, mv.img_dst, mv.tile_expert, (hipStream_t)stream);
            if (rc) return rc;
            // few rows per expert (on average at most 32): the int8 matrix-core kernel over 32-row chunks of the same expert-sorted image — the CPU's own integer
            // block dots (ggml_compute_forward_mul_mat_id's vec_dot calls, ggml-cpu.c:7752-7776) instead of fp16 tiles that would be mostly padding.  The int8
            // image (quants, d, bsums) takes the place of the fp16 one in the workspace.  Where it wins (MI355X, one box, 8 experts x 2 used x 4096^2, us per call at
            // 32 / 64 / 128 tokens, profiles/r04/moe64_ab.txt): Q6_K 95 / 96 / 125 vs 158 / 158 / 160 and Q8_0 91 / 92 / 133 vs 144 / 144 / 147 on the grouped per-lane-load
            // GEMM these formats have; NOT Q4_K, whose grouped k_gemm_kq_t64 takes 45 / 46 / 51 against 82 / 83 / 115 (CDNA4_MMQ_IDS=2 forces it there: 2e-7 from the
            // oracle instead of 3e-4).  CDNA4_NO_MMQ_IDS=1 turns the route off.
            static const bool no_mmq_ids = getenv("CDNA4_NO_MMQ_IDS") && atoi(getenv("CDNA4_NO_MMQ_IDS")) != 0;
            static const bool all_mmq_ids = getenv("CDNA4_MMQ_IDS") && atoi(getenv("CDNA4_MMQ_IDS")) == 2;
            if (!no_mmq_ids && (type != CDNA4_Q4_K || all_mmq_ids) && n_tok * n_used <= 32 * n_expert && cdna4_mmq_ids_supported(type, K) &&
                !(((uintptr_t)as | (uintptr_t)w_row_bytes | (uintptr_t)w_expert_bytes) & (type == CDNA4_Q6_K ? 1 : 15))) {
                int8_t *iq = (int8_t *)mv.xh;
                float *id_ = (float *)((uint8_t *)mv.xh + align256((size_t)mv.img_rows * K));
                int16_t *ibs = (int16_t *)((uint8_t *)id_ + align256((size_t)mv.img_rows * (K / 32) * 4));
                rc = is_kq(type) ? cdna4_launch_quantize_q8_K_gather_i8(b, b_row_stride, K, mv.img_rows, mv.img_src, iq, id_, ibs, (hipStream_t)stream)
                                 : cdna4_launch_quantize_q8_0_gather_i8(b, b_row_stride, K, mv.img_rows, mv.img_src, iq, id_, (hipStream_t)stream);
                if (rc) return rc;
                cdna4_gemv_args g{};
                g.type = type; g.W = (const uint8_t *)as; g.w_row_bytes = w_row_bytes; g.qs = iq; g.d = id_; g.bsums = is_kq(type) ? ibs : nullptr;
                g.Y = dst; g.y_col_stride = dst_row_stride; g.M = (int)M; g.K = (int)K; g.ncol = (int)mv.img_rows;
                return cdna4_launch_mmq_ids(g, mv.tile_expert, mv.img_dst, w_expert_bytes, (hipStream_t)stream);
            }
            rc = is_kq(type) ? cdna4_launch_quantize_q8_K_gather(b, b_row_stride, K, mv.img_rows, mv.img_src, mv.xh, (hipStream_t)stream)
                             : cdna4_launch_quantize_q8_0_gather(b, b_row_stride, K, mv.img_rows, mv.img_src, mv.xh, (hipStream_t)stream);
            if (rc) return rc;
            cdna4_gemm_args a{};
            a.type = type; a.W = (const uint8_t *)as; a.w_row_bytes = w_row_bytes; a.xh = mv.xh; a.xh_row_elems = K;
            a.Y = dst; a.y_row_elems = dst_row_stride; a.M = (int)M; a.K = (int)K; a.B = (int)mv.img_rows;
            // Q4_0 experts with a RESIDENT Q4_0R image of the whole stack (the experts' rows back to back: ggml's 3-D expert tensor): Q4_K's grouped kernel instead of the
            // per-lane-load one (round 5; the image is found by the stack's pointer, like ggml_cdna4_mul_mat finds a matrix's)
            if (type == CDNA4_Q4_0 && K % 256 == 0 && w_expert_bytes == M * w_row_bytes) {
                const uint8_t *img = cdna4_resident_lookup(CDNA4_Q4_0, as, w_row_bytes, M * n_expert, K);
                if (img && !((uintptr_t)img & 15)) {
                    a.type = CDNA4_Q4_0R; a.W = img; a.w_row_bytes = (K / 256) * 144;
                    return cdna4_launch_gemm_t64_ids(a, mv.tile_expert, mv.img_dst, M * a.w_row_bytes, (hipStream_t)stream);
                }
            }
            return cdna4_launch_gemm_ids(a, mv.tile_expert, mv.img_dst, w_expert_bytes, (hipStream_t)stream);   // Q4_K: k_gemm_kq_t64<.., IDS>; Q5_K / Q6_K / Q4_0 / Q8_0: k_gemm_q<.., IDS>
        }
    }
    if (n_tok == 1 && n_used <= 65535 && cdna4_gemv_fused_supported(type, K, 1) && !(((uintptr_t)b | (uintptr_t)(b_row_stride * 4)) & 15)) {
        // single-token decode of a mixture-of-experts layer: one launch, the activation quantizer runs inside the GEMV
        cdna4_gemv_args g{};
        g.type = type; g.W = (const uint8_t *)as; g.w_row_bytes = w_row_bytes; g.Y = dst; g.y_col_stride = dst_row_stride;
        g.M = (int)M; g.K = (int)K; g.ncol = (int)n_used;
        g.ids = ids; g.ids_tok_stride = ids_tok_stride; g.w_expert_bytes = w_expert_bytes; g.n_used = (int)n_used; g.n_b = (int)n_b; g.n_expert = (int)n_expert;
        return cdna4_launch_gemv_q_fused_ids(g, b, b_row_stride, (hipStream_t)stream);
    }
    if (!workspace || ((uintptr_t)workspace & 255)) return cdna4_set_error_msg("mul_mat_id: workspace must be 256-byte aligned");
    const ws_view v = carve(type, K, nact, workspace);
    if (workspace_bytes < v.total) return cdna4_set_error_msg("mul_mat_id: workspace too small");
    int rc = ggml_cdna4_prepare_act(type, b, b_row_stride, K, nact, workspace, workspace_bytes, GGML_CDNA4_PATH_GEMV, stream);
    if (rc) return rc;
    cdna4_gemv_args g{};
    g.type = type; g.W = (const uint8_t *)as; g.w_row_bytes = w_row_bytes; g.qs = v.qs; g.d = v.d; g.bsums = v.bsums;
    g.Y = dst; g.y_col_stride = dst_row_stride; g.M = (int)M; g.K = (int)K; g.ncol = (int)(n_tok * n_used);
    g.ids = ids; g.ids_tok_stride = ids_tok_stride; g.w_expert_bytes = w_expert_bytes; g.n_used = (int)n_used; g.n_b = (int)n_b; g.n_expert = (int)n_expert;
    return cdna4_launch_gemv_q(g, (hipStream_t)stream);
}

}  // extern "C"
