// ggml_cdna4_backend.cpp — the ggml backend plug-in for MI355X: implements the five function-pointer tables of
// ggml's backend ABI (src/ggml-backend-impl.h:17-207) and exports `ggml_backend_init` (:215,222-228), so the
// UNMODIFIED reference binaries load it with GGML_BACKEND_PATH=<this .so> (src/ggml-backend-reg.cpp:577-581)
// or ggml_backend_load(path).  It is a thin strides-to-pointers layer: all arithmetic is in libcdna4_kernels.so
// (C-ABI include/ggml_cdna4.h + the op table include/ggml_cdna4_ops.h).  Written from scratch against the ABI;
// it shares no code with src/ggml-cuda or the src/ggml-hip shim.
//
// Error conventions follow SURVEY.md §8(b): programmer errors abort (GGML_ASSERT), OOM -> NULL buffer,
// runtime failures -> GGML_STATUS_FAILED.  Unsupported ops are declined in supports_op (the scheduler then
// places them on the CPU backend); nothing is ever computed on the host here.
#include "ggml_cdna4_internal.h"
#include "ggml_cdna4_ops.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

struct cdna4_device_ctx { int device; std::string name, description; };
struct cdna4_buft_ctx   { int device; std::string name; bool resident = false; };
// resident: the buffer keeps, beside every eligible weight matrix, its kernel-native image (ggml_cdna4_resident_image_*): see the CDNA4_Resident buffer type below
// Everything the image needs is kept BY VALUE: the ggml contract lets the ggml_context (and with it every ggml_tensor) be freed before the buffer — the reference's own gpt-2
// program does (examples/gpt-2/main-backend.cpp:936-939: ggml_free(ctx_w), then ggml_backend_buffer_free(buffer_w)) — so nothing here may touch the tensor after init_tensor
// (the first version kept the ggml_tensor pointer and read ->data / ->ne in free_buffer).
struct cdna4_resident_tensor {
    void * data; int type; int64_t ne0, rows, nb1; size_t nbytes; std::string name;      // the weight: address, ggml type, K, ne[1] * ne[2], row stride, byte size
    void * image; size_t written; bool registered;
    bool overlaps(const void * p, size_t n) const { return (const char *)p < (const char *)data + nbytes && (const char *)data < (const char *)p + n; }
};
struct cdna4_buffer_ctx { int device; void * base; size_t size; bool resident = false; std::vector<cdna4_resident_tensor> res; bool host_registered = false; };

static ggml_backend_reg_t ggml_backend_cdna4_reg(void);
static ggml_backend_buffer_type_t cdna4_buffer_type(int device);
static const char * cdna4_buffer_get_name_tag = "CDNA4";

// ============================================================================================================
// buffer
static void cdna4_buffer_free(ggml_backend_buffer_t buffer) {
    cdna4_buffer_ctx * ctx = (cdna4_buffer_ctx *)buffer->context;
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipDeviceSynchronize());
    for (cdna4_resident_tensor & r : ctx->res) {
        if (r.registered) (void)ggml_cdna4_resident_image_unregister(r.data);
        if (r.image) HIP_OK(hipFree(r.image));
    }
    if (ctx->host_registered) { HIP_OK(hipHostUnregister(ctx->base)); delete ctx; return; }      // (buffer_from_host_ptr: the memory is the host's)
    HIP_OK(hipFree(ctx->base));
    delete ctx;
}
static void * cdna4_buffer_get_base(ggml_backend_buffer_t buffer) { return ((cdna4_buffer_ctx *)buffer->context)->base; }
static bool buffer_is_cdna4(ggml_backend_buffer_t buffer) { return buffer && buffer->iface.get_base == cdna4_buffer_get_base; }

static void resident_written(cdna4_buffer_ctx * ctx, ggml_tensor * tensor, size_t offset, size_t size);      // (below: CDNA4_Resident)
static void cdna4_buffer_memset_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    cdna4_buffer_ctx * ctx = (cdna4_buffer_ctx *)buffer->context;
    HIP_OK(hipSetDevice(ctx->device));
    if (ctx->host_registered) { HIP_OK(hipDeviceSynchronize()); memset((char *)tensor->data + offset, value, size); return; }      // (the host's own memory)
    HIP_OK(hipMemset((char *)tensor->data + offset, value, size));
    HIP_OK(hipStreamSynchronize(0));
    resident_written(ctx, tensor, offset, size);
}
// ---- CDNA4_Resident: a weight written into such a buffer gets its kernel-native image built ONCE, when its last byte has arrived (set_tensor / cpy_tensor: the
// reference repacks at the same moment, src/ggml-cpu/ggml-cpu-aarch64.cpp:4144-4172) — built twice and compared (a re-encoding that is not bit-stable is refused: the tensor
// then simply has no image and every call re-encodes as before).  The original bytes stay in the buffer: get_tensor is the plain copy, decode reads them, views work.
// the weight whose bytes [p, p + n) touches: matched by ADDRESS RANGE — a set_tensor / memset / cpy on a view at a non-zero offset (a row range of a
// weight) or a graph node writing into a weight must find the image too, or prefill (the image) and decode (the source bytes) silently disagree
static cdna4_resident_tensor * resident_find(cdna4_buffer_ctx * ctx, const void * p, size_t n) {
    for (cdna4_resident_tensor & r : ctx->res) if (r.overlaps(p, n ? n : 1)) return &r;
    return nullptr;
}
static void resident_invalidate(cdna4_resident_tensor * r) {
    if (r->registered) { (void)ggml_cdna4_resident_image_unregister(r->data); r->registered = false; }
}
// bytes [p, p + n) of the buffer have been rewritten
static void resident_written_range(cdna4_buffer_ctx * ctx, const void * p, size_t n) {
    if (!ctx->resident || n == 0) return;
    for (cdna4_resident_tensor & r : ctx->res) {
        if (!r.overlaps(p, n)) continue;
        resident_invalidate(&r);
        const char * lo = (const char *)p > (const char *)r.data ? (const char *)p : (const char *)r.data;
        const char * hi = (const char *)p + n < (const char *)r.data + r.nbytes ? (const char *)p + n : (const char *)r.data + r.nbytes;
        const size_t got = (size_t)(hi - lo);
        r.written = (lo == (const char *)r.data && got >= r.nbytes) ? r.nbytes : r.written + got;       // (loaders write whole tensors; pieces are counted until they add up)
        if (r.written < r.nbytes) continue;
        r.written = 0;
        // (rows = ne[1] * ne[2]: a contiguous expert stack is one image, found by the stack's pointer in MUL_MAT_ID)
        if (ggml_cdna4_resident_image_register(r.type, r.data, r.nb1, r.rows, r.ne0, r.image, 1, nullptr) == 0) r.registered = true;
        else fprintf(stderr, "ggml-cdna4: no resident image for %s: %s\n", r.name.c_str(), ggml_cdna4_last_error());
    }
}
static void resident_written(cdna4_buffer_ctx * ctx, ggml_tensor * tensor, size_t offset, size_t size) { resident_written_range(ctx, (const char *)tensor->data + offset, size); }
// a graph node wrote [p, p + n) (run_nodes): a weight underneath loses its image until it is written whole again (CPY / SET_ROWS into a weight, a quantized KV tensor
// that was placed in this buffer type) — the per-call routes then read the source bytes, as the decode kernels always do
static void cdna4_resident_node_wrote(ggml_backend_buffer_t buffer, const void * p, size_t n) {
    cdna4_buffer_ctx * ctx = (cdna4_buffer_ctx *)buffer->context;
    if (!ctx->resident) return;
    for (cdna4_resident_tensor & r : ctx->res) if (r.registered && r.overlaps(p, n)) { resident_invalidate(&r); r.written = 0; }
}
static void cdna4_resident_init_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor) {
    cdna4_buffer_ctx * ctx = (cdna4_buffer_ctx *)buffer->context;
    // 2-D weights; 3-D expert stacks only for Q4_0, the one format whose grouped MUL_MAT_ID reads an image (capi.hip: ggml_cdna4_mul_mat_id)
    if (tensor->view_src || tensor->ne[3] != 1 || !ggml_is_contiguous(tensor) || (tensor->ne[2] != 1 && tensor->type != GGML_TYPE_Q4_0)) return;
    const size_t bytes = ggml_cdna4_resident_image_size((int)tensor->type, tensor->ne[1] * tensor->ne[2], tensor->ne[0]);
    if (bytes == 0) return;                                                         // a type that needs no image (or no MUL_MAT weight at all)
    if (resident_find(ctx, tensor->data, ggml_nbytes(tensor))) return;
    HIP_OK(hipSetDevice(ctx->device));
    void * img = nullptr;
    if (hipMalloc(&img, bytes) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "ggml-cdna4: no memory for the resident image of %s (per-call re-encoding stays)\n", tensor->name); return; }
    ctx->res.push_back(cdna4_resident_tensor{tensor->data, (int)tensor->type, tensor->ne[0], tensor->ne[1] * tensor->ne[2], (int64_t)tensor->nb[1], ggml_nbytes(tensor), tensor->name, img, 0, false});
}
static void cdna4_buffer_set_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    cdna4_buffer_ctx * ctx = (cdna4_buffer_ctx *)buffer->context;
    HIP_OK(hipSetDevice(ctx->device));
    if (ctx->host_registered) { HIP_OK(hipDeviceSynchronize()); memcpy((char *)tensor->data + offset, data, size); return; }
    HIP_OK(hipMemcpy((char *)tensor->data + offset, data, size, hipMemcpyHostToDevice));
    resident_written(ctx, tensor, offset, size);
}
static void cdna4_buffer_get_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    cdna4_buffer_ctx * ctx = (cdna4_buffer_ctx *)buffer->context;
    HIP_OK(hipSetDevice(ctx->device));
    // (no device-wide synchronize: a blocking copy on the null stream is ordered behind everything the backend's stream — a
    //  blocking stream — was given before this call)
    if (ctx->host_registered) { HIP_OK(hipDeviceSynchronize()); memcpy(data, (const char *)tensor->data + offset, size); return; }
    HIP_OK(hipMemcpy(data, (const char *)tensor->data + offset, size, hipMemcpyDeviceToHost));
}
static bool cdna4_buffer_cpy_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * src, ggml_tensor * dst) {
    if (!buffer_is_cdna4(src->buffer)) return false;
    cdna4_buffer_ctx * sctx = (cdna4_buffer_ctx *)src->buffer->context;
    cdna4_buffer_ctx * dctx = (cdna4_buffer_ctx *)buffer->context;
    HIP_OK(hipSetDevice(sctx->device)); HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipSetDevice(dctx->device));
    if (sctx->host_registered || dctx->host_registered) HIP_OK(hipMemcpy(dst->data, src->data, ggml_nbytes(src), hipMemcpyDefault));
    else if (sctx->device == dctx->device) HIP_OK(hipMemcpy(dst->data, src->data, ggml_nbytes(src), hipMemcpyDeviceToDevice));
    else HIP_OK(hipMemcpyPeer(dst->data, dctx->device, src->data, sctx->device, ggml_nbytes(src)));
    HIP_OK(hipDeviceSynchronize());
    resident_written(dctx, dst, 0, ggml_nbytes(src));
    return true;
}
static void cdna4_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    cdna4_buffer_ctx * ctx = (cdna4_buffer_ctx *)buffer->context;
    HIP_OK(hipSetDevice(ctx->device));
    if (ctx->host_registered) { HIP_OK(hipDeviceSynchronize()); memset(ctx->base, value, ctx->size); return; }
    HIP_OK(hipMemset(ctx->base, value, ctx->size));
    HIP_OK(hipDeviceSynchronize());
    for (cdna4_resident_tensor & r : ctx->res) { resident_invalidate(&r); r.written = 0; }       // (the images follow the next whole write)
}
static const ggml_backend_buffer_i cdna4_buffer_iface = {
    /* .free_buffer   = */ cdna4_buffer_free,
    /* .get_base      = */ cdna4_buffer_get_base,
    /* .init_tensor   = */ NULL,
    /* .memset_tensor = */ cdna4_buffer_memset_tensor,
    /* .set_tensor    = */ cdna4_buffer_set_tensor,
    /* .get_tensor    = */ cdna4_buffer_get_tensor,
    /* .cpy_tensor    = */ cdna4_buffer_cpy_tensor,
    /* .clear         = */ cdna4_buffer_clear,
    /* .reset         = */ NULL,
};
static const ggml_backend_buffer_i cdna4_resident_buffer_iface = {
    /* .free_buffer   = */ cdna4_buffer_free,
    /* .get_base      = */ cdna4_buffer_get_base,
    /* .init_tensor   = */ cdna4_resident_init_tensor,
    /* .memset_tensor = */ cdna4_buffer_memset_tensor,
    /* .set_tensor    = */ cdna4_buffer_set_tensor,
    /* .get_tensor    = */ cdna4_buffer_get_tensor,
    /* .cpy_tensor    = */ cdna4_buffer_cpy_tensor,
    /* .clear         = */ cdna4_buffer_clear,
    /* .reset         = */ NULL,
};

// ============================================================================================================
// buffer type
static const char * cdna4_buft_get_name(ggml_backend_buffer_type_t buft) { return ((cdna4_buft_ctx *)buft->context)->name.c_str(); }
static ggml_backend_buffer_t cdna4_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    cdna4_buft_ctx * bctx = (cdna4_buft_ctx *)buft->context;
    if (hipSetDevice(bctx->device) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    void * base = nullptr;
    const size_t alloc = size + 256;                       // slack: kernels may read whole 16-byte pieces
    if (hipMalloc(&base, alloc) != hipSuccess) {           // OOM -> NULL, like ggml-cuda.cu:646-651
        (void)hipGetLastError();
        fprintf(stderr, "ggml-cdna4: allocating %.2f MiB on device %d failed\n", size / 1048576.0, bctx->device);
        return NULL;
    }
    cdna4_buffer_ctx * ctx = new cdna4_buffer_ctx{bctx->device, base, alloc};
    ctx->resident = bctx->resident;
    return ggml_backend_buffer_init(buft, bctx->resident ? cdna4_resident_buffer_iface : cdna4_buffer_iface, ctx, size);
}
static size_t cdna4_buft_get_alignment(ggml_backend_buffer_type_t) { return 256; }
static size_t cdna4_buft_get_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * tensor) {
    return (ggml_nbytes(tensor) + 15) & ~(size_t)15;       // 16-byte granules: vector loads never straddle tensors
}
static bool cdna4_buft_is_host(ggml_backend_buffer_type_t) { return false; }
static const ggml_backend_buffer_type_i cdna4_buft_iface = {
    /* .get_name       = */ cdna4_buft_get_name,
    /* .alloc_buffer   = */ cdna4_buft_alloc_buffer,
    /* .get_alignment  = */ cdna4_buft_get_alignment,
    /* .get_max_size   = */ NULL,
    /* .get_alloc_size = */ cdna4_buft_get_alloc_size,
    /* .is_host        = */ cdna4_buft_is_host,
};
static bool buft_is_cdna4(ggml_backend_buffer_type_t buft) { return buft && buft->iface.get_name == cdna4_buft_get_name; }

// ============================================================================================================
// op support + dispatch
// every type here has int8-dot GEMV units and an MFMA prefill path (its own kernel, or an exact re-encoding into a format that has one: ggml_cdna4.h)
static bool is_qweight(ggml_type t) { return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K ||
                                             t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K ||
                                             t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_IQ4_NL || t == GGML_TYPE_IQ4_XS; }

static bool supports_mul_mat(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    if (op->type != GGML_TYPE_F32 || !ggml_is_contiguous(op)) return false;
    if (a->buffer && cdna4_buft_is_split(a->buffer->buft)) return cdna4_split_supports_mul_mat(op);
    if (is_qweight(a->type)) {
        // src1 must be F32 (the CPU backend itself only takes F32 or vec_dot_type, src/ggml-cpu/ggml-cpu.cpp:404-405)
        if (b->type != GGML_TYPE_F32) return false;
        if (a->nb[0] != ggml_type_size(a->type) || b->nb[0] != sizeof(float)) return false;
        if (b->nb[1] % 16 || b->nb[2] % 16 || b->nb[3] % 16) return false;
        return true;
    }
    return cdna4_ops_supports_matmul(op);
}

static bool supports_mul_mat_id(const ggml_tensor * op) {
    const ggml_tensor * as = op->src[0], * b = op->src[1], * ids = op->src[2];
    if (!is_qweight(as->type) || b->type != GGML_TYPE_F32 || ids->type != GGML_TYPE_I32 || op->type != GGML_TYPE_F32) return false;
    if (!ggml_is_contiguous(as) || !ggml_is_contiguous(b) || !ggml_is_contiguous(op)) return false;
    if (ids->nb[0] != sizeof(int32_t) || as->ne[3] != 1 || b->ne[3] != 1) return false;
    // what ggml_cdna4_mul_mat_id accepts: any b->ne[1] dividing n_used (slot u reads row u % ne11, as ggml-cpu.c:7752), 16-byte
    // activation rows (the quantizers load float4)
    if (b->ne[1] <= 0 || ids->ne[0] % b->ne[1] || b->nb[1] % 16) return false;
    return true;
}

// ---- hand-off of quantized activations between MUL_MATs of the same src1 (wq / wk / wv, w_gate / w_up).  The kernel library leaves the image of a
// call's activations in the workspace and says which (ggml_cdna4_act_image_key); the graph walk knows whether src1 and the workspace are untouched since.  The second
// product is then ggml_cdna4_mul_mat_prepared[_fused]: the same kernel on the same image (bit-identical), one launch fewer.  GGML_CDNA4_NO_ACT_SHARE=1: off.
static std::atomic<int> g_act_shared{0};
extern "C" int ggml_backend_cdna4_act_shared_count(void) { return g_act_shared.load(); }
static bool act_share_on() { static const bool off = getenv("GGML_CDNA4_NO_ACT_SHARE") != nullptr; return !off && !cdna4_exact_mode(); }
// does the workspace hold the image a MUL_MAT of (type, M, K, B) over X would build?
static bool act_image_ready(const cdna4_backend_ctx * ctx, const ggml_tensor * a, int64_t M, int64_t K, int64_t B, const void * X, int64_t x_stride, size_t need) {
    const auto & im = ctx->act_image;
    if (!act_share_on() || !im.key || im.uses != ctx->ws_uses || im.x != X || im.x_stride != x_stride || im.K != K || im.B != B || need > ctx->ws_size) return false;
    return ggml_cdna4_act_image_key_of((int)a->type, a->data, (int64_t)a->nb[1], M, K, B) == im.key;
}
// a ggml_cdna4_mul_mat[_fused] call of weight a (M x K) with B rows of X has just been issued on the workspace.  The key is the CONCRETE matrix's (its alignment decides
// whether the few-row call wrote the int8 image or quantized inside a one-launch GEMV and left the workspace alone: ADVICE r5)
static void act_image_note(cdna4_backend_ctx * ctx, const ggml_tensor * a, int64_t M, int64_t K, int64_t B, const void * X, int64_t x_stride) {
    auto & im = ctx->act_image;
    im.key = act_share_on() ? ggml_cdna4_act_image_key_of((int)a->type, a->data, (int64_t)a->nb[1], M, K, B) : 0;
    im.x = X; im.x_stride = x_stride; im.K = K; im.B = B; im.x_bytes = (size_t)((B - 1) * x_stride + K) * sizeof(float); im.uses = ctx->ws_uses;
}
// a node has written [p, p + n): activations that overlap it are no longer the ones the image was made of
static void act_image_written(cdna4_backend_ctx * ctx, const void * p, size_t n) {
    auto & im = ctx->act_image;
    if (im.key && (const char *)p < (const char *)im.x + im.x_bytes && (const char *)im.x < (const char *)p + n) im.key = 0;
}
// ... and a weight of a CDNA4_Resident buffer underneath it loses its kernel-native image (CPY / SET_ROWS into a weight, a quantized KV tensor placed there)
static void node_wrote(cdna4_backend_ctx * ctx, const ggml_tensor * t) {
    act_image_written(ctx, t->data, ggml_nbytes(t));
    {   // ... and a MUL_MAT_ID front made of activations / ids that overlap it
        auto & mf = ctx->moe_front;
        const char * p = (const char *)t->data; const size_t n = ggml_nbytes(t);
        if (mf.key && ((p < (const char *)mf.b + mf.b_bytes && (const char *)mf.b < p + n) || (p < (const char *)mf.ids + mf.ids_bytes && (const char *)mf.ids < p + n))) mf.key = 0;
    }
    ggml_backend_buffer_t buf = t->view_src ? t->view_src->buffer : t->buffer;
    if (buffer_is_cdna4(buf) && ((cdna4_buffer_ctx *)buf->context)->resident) cdna4_resident_node_wrote(buf, t->data, ggml_nbytes(t));
}

static enum ggml_status compute_mul_mat(cdna4_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * a = dst->src[0], * b = dst->src[1];
    if (a->buffer && cdna4_buft_is_split(a->buffer->buft)) return cdna4_split_mul_mat(ctx, dst);
    if (!is_qweight(a->type)) return cdna4_ops_compute(ctx, dst);
    const int64_t K = a->ne[0], M = a->ne[1], N = b->ne[1];
    const int64_t r2 = b->ne[2] / a->ne[2], r3 = b->ne[3] / a->ne[3];
    const bool exact = cdna4_exact_mode() && ggml_cdna4_mul_mat_exact_supported((int)a->type, K);      // GGML_CDNA4_EXACT=1: the CPU backend's own summation order (exact.hip)
    // all activation rows form one matrix when src0 has no batch dims and src1's batch dims are laid out row after row
    const bool collapse = a->ne[2] == 1 && a->ne[3] == 1 && b->nb[2] == (size_t)N * b->nb[1] && b->nb[3] == (size_t)b->ne[2] * b->nb[2];
    const int64_t nbatch = collapse ? 1 : b->ne[2] * b->ne[3];
    const int64_t Bc = collapse ? N * b->ne[2] * b->ne[3] : N;
    const size_t need = exact ? ggml_cdna4_mul_mat_exact_workspace_size((int)a->type, K, Bc) : ggml_cdna4_mul_mat_workspace_size((int)a->type, K, Bc);
    if (nbatch == 1 && !exact && act_image_ready(ctx, a, M, K, Bc, b->data, (int64_t)(b->nb[1] / sizeof(float)), need)) {
        // the previous MUL_MAT quantized these very activations into the workspace: multiply them (the image stays valid for the next reader)
        if (ggml_cdna4_mul_mat_prepared((int)a->type, a->data, (int64_t)a->nb[1], (float *)dst->data, (int64_t)(dst->nb[1] / sizeof(float)), M, K, Bc, ctx->ws, ctx->ws_size,
                                        GGML_CDNA4_PATH_AUTO, 0, 0, ctx->stream) == 0) { ctx->n_act_shared++; g_act_shared++; return GGML_STATUS_SUCCESS; }
        ctx->act_image.key = 0;                                         // (refused before any launch: the plain call below quantizes again — ADVICE r5)
    }
    void * ws = ctx->need_ws(need);
    if (!ws) return GGML_STATUS_ALLOC_FAILED;
    for (int64_t ib = 0; ib < nbatch; ib++) {
        const int64_t i12 = collapse ? 0 : ib % b->ne[2], i13 = collapse ? 0 : ib / b->ne[2];
        const char * W = (const char *)a->data + (i12 / r2) * a->nb[2] + (i13 / r3) * a->nb[3];
        const float * X = (const float *)((const char *)b->data + i12 * b->nb[2] + i13 * b->nb[3]);
        float * Y = (float *)((char *)dst->data + i12 * dst->nb[2] + i13 * dst->nb[3]);
        const int rc = exact ? ggml_cdna4_mul_mat_exact((int)a->type, W, (int64_t)a->nb[1], X, (int64_t)(b->nb[1] / sizeof(float)), Y, (int64_t)(dst->nb[1] / sizeof(float)),
                                                        M, K, Bc, ws, ctx->ws_size, ctx->stream)
                             : ggml_cdna4_mul_mat((int)a->type, W, (int64_t)a->nb[1], X, (int64_t)(b->nb[1] / sizeof(float)), Y, (int64_t)(dst->nb[1] / sizeof(float)),
                                                  M, K, Bc, ws, ctx->ws_size, GGML_CDNA4_PATH_AUTO, 0, 0, ctx->stream);
        if (rc) { fprintf(stderr, "ggml-cdna4: MUL_MAT failed: %s\n", ggml_cdna4_last_error()); return GGML_STATUS_FAILED; }
    }
    if (nbatch == 1 && !exact) act_image_note(ctx, a, M, K, Bc, b->data, (int64_t)(b->nb[1] / sizeof(float)));
    return GGML_STATUS_SUCCESS;
}

static std::atomic<int> g_moe_front_shared{0};
extern "C" int ggml_backend_cdna4_moe_front_shared_count(void) { return g_moe_front_shared.load(); }
static enum ggml_status compute_mul_mat_id(cdna4_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * as = dst->src[0], * b = dst->src[1], * ids = dst->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], n_expert = as->ne[2];
    const int64_t n_b = b->ne[1], n_tok = b->ne[2], n_used = ids->ne[0];
    const int64_t b_row = (int64_t)(b->nb[1] / 4), b_tok = (int64_t)(b->nb[2] / 4), ids_tok = (int64_t)(ids->nb[1] / 4);
    const size_t need = ggml_cdna4_mul_mat_id_workspace_size((int)as->type, K, n_expert, n_used, n_b, n_tok);
    auto & mf = ctx->moe_front;
    // the previous MUL_MAT_ID sorted these very ids and quantized these very activations into the workspace (w_up, then w_gate, of a mixture-of-experts layer): multiply its front
    if (act_share_on() && mf.key && mf.uses == ctx->ws_uses && need <= ctx->ws_size && mf.b == b->data && mf.ids == ids->data && mf.type == (int)as->type && mf.b_row == b_row &&
        mf.b_tok == b_tok && mf.ids_tok == ids_tok && mf.M == M && mf.K == K && mf.n_expert == n_expert && mf.n_used == n_used && mf.n_b == n_b && mf.n_tok == n_tok &&
        ggml_cdna4_mul_mat_id_front_key((int)as->type, as->data, (int64_t)as->nb[1], (int64_t)as->nb[2], M, K, n_expert, n_used, n_b, n_tok, ctx->ws_size) == mf.key) {
        const int rc = ggml_cdna4_mul_mat_id_prepared((int)as->type, as->data, (int64_t)as->nb[1], (int64_t)as->nb[2], (const float *)b->data, b_row, b_tok, (const int32_t *)ids->data, ids_tok,
                                                      (float *)dst->data, (int64_t)(dst->nb[1] / 4), (int64_t)(dst->nb[2] / 4), M, K, n_expert, n_used, n_b, n_tok, ctx->ws, ctx->ws_size, ctx->stream);
        if (rc == 0) { ctx->n_moe_front_shared++; g_moe_front_shared++; return GGML_STATUS_SUCCESS; }
        mf.key = 0;                                                     // (refused before any launch: the full call below)
    }
    void * ws = ctx->need_ws(need);
    if (!ws) return GGML_STATUS_ALLOC_FAILED;
    const int rc = ggml_cdna4_mul_mat_id((int)as->type, as->data, (int64_t)as->nb[1], (int64_t)as->nb[2],
                                         (const float *)b->data, b_row, b_tok,
                                         (const int32_t *)ids->data, ids_tok,
                                         (float *)dst->data, (int64_t)(dst->nb[1] / 4), (int64_t)(dst->nb[2] / 4),
                                         M, K, n_expert, n_used, n_b, n_tok, ws, ctx->ws_size, ctx->stream);
    if (rc) { fprintf(stderr, "ggml-cdna4: MUL_MAT_ID failed: %s\n", ggml_cdna4_last_error()); return GGML_STATUS_FAILED; }
    mf.key = act_share_on() ? ggml_cdna4_mul_mat_id_front_key((int)as->type, as->data, (int64_t)as->nb[1], (int64_t)as->nb[2], M, K, n_expert, n_used, n_b, n_tok, ctx->ws_size) : 0;
    mf.uses = ctx->ws_uses; mf.b = b->data; mf.ids = ids->data; mf.type = (int)as->type; mf.b_row = b_row; mf.b_tok = b_tok; mf.ids_tok = ids_tok;
    mf.b_bytes = (size_t)((n_tok - 1) * b_tok + (n_b - 1) * b_row + K) * 4; mf.ids_bytes = (size_t)((n_tok - 1) * ids_tok + n_used) * 4;
    mf.M = M; mf.K = K; mf.n_expert = n_expert; mf.n_used = n_used; mf.n_b = n_b; mf.n_tok = n_tok;
    return GGML_STATUS_SUCCESS;
}

// ============================================================================================================
// backend (stream)
static const char * cdna4_backend_get_name(ggml_backend_t backend) { return ((cdna4_backend_ctx *)backend->context)->name.c_str(); }
static void cdna4_backend_free(ggml_backend_t backend) {
    cdna4_backend_ctx * ctx = (cdna4_backend_ctx *)backend->context;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    cdna4_split_free_lanes(ctx);
    if (getenv("GGML_CDNA4_STATS")) fprintf(stderr, "ggml-cdna4: %s: %d HIP-graph captures, %d replays\n", ctx->name.c_str(), ctx->n_graph_captures, ctx->n_graph_launches);
    if (getenv("GGML_CDNA4_STATS") && ctx->n_grouped) fprintf(stderr, "ggml-cdna4: %s: %d one-row MUL_MATs rode in the launch of another product of the same activations\n", ctx->name.c_str(), ctx->n_grouped);
    if (getenv("GGML_CDNA4_STATS") && ctx->n_act_shared) fprintf(stderr, "ggml-cdna4: %s: %d MUL_MATs multiplied quantized activations that were already in the workspace (%d images left by NORM chains)\n", ctx->name.c_str(), ctx->n_act_shared, ctx->n_act_produced);
    if (getenv("GGML_CDNA4_STATS") && (ctx->n_ksplit_rccl || ctx->n_ksplit_sum)) fprintf(stderr, "ggml-cdna4: %s: K-split MUL_MAT: %d RCCL all-reduces, %d in-order sums\n", ctx->name.c_str(), ctx->n_ksplit_rccl, ctx->n_ksplit_sum);
    for (auto & gs : ctx->graph_slots) if (gs.exec) (void)hipGraphExecDestroy(gs.exec);
    if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
    if (ctx->ws) (void)hipFree(ctx->ws);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    delete backend;
}
static void cdna4_backend_synchronize(ggml_backend_t backend) {
    cdna4_backend_ctx * ctx = (cdna4_backend_ctx *)backend->context;
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    // (synchronize has no status to return: the fault stays pending for the next graph_compute, which fails; here it is only announced)
    if (const int fault = ggml_cdna4_device_fault(0)) fprintf(stderr, "ggml-cdna4: %s: a launch gave up waiting for a co-resident work-group (code %d): the results behind this synchronize hold NaN tiles\n", ctx->name.c_str(), fault);
}
// ---- graph peepholes: chains every transformer graph contains, run as one launch each (bit-identical to the node-by-node sequence:
// the fused kernels perform the same separate fp32 operations — include/ggml_cdna4.h).  A chain is taken only if every intermediate
// result has exactly one reader in this graph (the next link) and is not flagged as a graph output; the last node's tensor receives
// the result, the intermediates are never written.  GGML_CDNA4_NO_FUSE=1 turns all of it off (tests compare both).
//   MUL_MAT(quantized) -> ADD(bias) [-> GELU | -> ADD(residual)]     gpt-2: main-backend.cpp:515-521, 595-600, 656-666, 690-698
//   NORM | RMS_NORM -> MUL(gain) [-> ADD(shift)]                      main-backend.cpp:476-488, 610-622
//   SCALE -> DIAG_MASK_INF -> SOFT_MAX                                main-backend.cpp:586-596
struct use_counts {
    std::unordered_map<const ggml_tensor *, int> n;
    use_counts(std::nullptr_t, int) {}
    explicit use_counts(ggml_cgraph * g) {
        for (int i = 0; i < ggml_graph_n_nodes(g); i++) {
            const ggml_tensor * t = ggml_graph_node(g, i);
            for (int j = 0; j < GGML_MAX_SRC; j++) if (t->src[j]) n[t->src[j]]++;
            if (t->view_src) n[t->view_src]++;
        }
    }
    bool only_reader(const ggml_tensor * t, const ggml_tensor * reader) const {          // t feeds `reader` and nothing else
        if (t->flags & GGML_TENSOR_FLAG_OUTPUT) return false;
        auto it = n.find(t);
        if (it == n.end() || it->second != 1) return false;
        for (int j = 0; j < GGML_MAX_SRC; j++) if (reader->src[j] == t) return true;
        return false;
    }
};
static bool is_row_vector_f32(const ggml_tensor * t, int64_t n) {
    return t && t->type == GGML_TYPE_F32 && t->ne[0] == n && t->ne[1] == 1 && t->ne[2] == 1 && t->ne[3] == 1 && t->nb[0] == sizeof(float) && t->data;
}
// The graph allocator hands the memory of a tensor whose last reader has been scheduled to later nodes — node by node that is safe; a
// fused launch reads the chain's INPUT while it writes the chain's LAST node, so the two must not share memory (an exact alias is fine for
// the row-local kernels — each element / row is read by the thread that writes it — never for MUL_MAT, whose every output reads all of X).
static bool mem_overlap(const ggml_tensor * a, const ggml_tensor * b) {
    const char * pa = (const char *)a->data, * pb = (const char *)b->data;
    return pa < pb + ggml_nbytes(b) && pb < pa + ggml_nbytes(a);
}
static bool alias_or_disjoint(const ggml_tensor * in, const ggml_tensor * out) { return in->data == out->data || !mem_overlap(in, out); }
static const ggml_tensor * other_src(const ggml_tensor * op, const ggml_tensor * t) { return op->src[0] == t ? op->src[1] : (op->src[1] == t ? op->src[0] : nullptr); }

// MUL_MAT at node i with its tail; returns the number of nodes consumed (0 = no chain here)
static int try_fused_mul_mat(cdna4_backend_ctx * ctx, ggml_cgraph * g, int i, const use_counts & uses, enum ggml_status & st) {
    ggml_tensor * mm = ggml_graph_node(g, i);
    const ggml_tensor * a = mm->src[0], * b = mm->src[1];
    const int n_nodes = ggml_graph_n_nodes(g);
    if (!is_qweight(a->type) || (a->buffer && cdna4_buft_is_split(a->buffer->buft)) || i + 1 >= n_nodes) return 0;
    if (a->ne[2] != 1 || a->ne[3] != 1 || b->ne[2] != 1 || b->ne[3] != 1 || !ggml_is_contiguous(mm)) return 0;    // plain 2-D products
    const int64_t M = a->ne[1], K = a->ne[0], B = b->ne[1];
    ggml_tensor * n1 = ggml_graph_node(g, i + 1);
    if (n1->op != GGML_OP_ADD || !uses.only_reader(mm, n1) || !ggml_are_same_shape(n1, mm) || !ggml_is_contiguous(n1)) return 0;
    const ggml_tensor * bias = other_src(n1, mm);
    if (!is_row_vector_f32(bias, M)) return 0;
    ggml_tensor * last = n1; int used = 2, act = 0; const ggml_tensor * resid = nullptr;
    if (i + 2 < n_nodes) {
        ggml_tensor * n2 = ggml_graph_node(g, i + 2);
        if (uses.only_reader(n1, n2) && ggml_are_same_shape(n2, n1) && ggml_is_contiguous(n2)) {
            if (n2->op == GGML_OP_UNARY && ggml_get_unary_op(n2) == GGML_UNARY_OP_GELU && n2->src[0] == n1) { act = 1; last = n2; used = 3; }
            else if (n2->op == GGML_OP_ADD) {
                const ggml_tensor * r = other_src(n2, n1);
                if (r && r != n1 && r->type == GGML_TYPE_F32 && ggml_are_same_shape(r, n1) && r->nb[0] == sizeof(float) && r->nb[1] % sizeof(float) == 0 && r->data) { resid = r; last = n2; used = 3; }
            }
        }
    }
    if (mem_overlap(last, b) || mem_overlap(last, a) || mem_overlap(last, bias)) return 0;       // (see mem_overlap)
    // the residual may BE the output (ggml_add_inplace, or the graph allocator placing ADD(resid, cur) onto resid) only where the tail is applied
    // in the store that produces the element — ggml_cdna4_mul_mat_fused_residual_may_alias says for which calls that holds
    if (resid && (ggml_cdna4_mul_mat_fused_residual_may_alias((int)a->type, M, K, B) ? !alias_or_disjoint(resid, last) : mem_overlap(resid, last))) return 0;
    // an aliased residual must be the SAME view (same row stride): the kernel library refuses any other overlap, and a refusal here would fail a graph
    // whose nodes run fine one by one
    if (resid && resid->data == last->data && resid->nb[1] != last->nb[1]) return 0;
    const size_t need = ggml_cdna4_mul_mat_workspace_size((int)a->type, K, B);
    if (act_image_ready(ctx, a, M, K, B, b->data, (int64_t)(b->nb[1] / sizeof(float)), need)) {         // (compute_mul_mat: the hand-off of quantized activations)
        if (ggml_cdna4_mul_mat_prepared_fused((int)a->type, a->data, (int64_t)a->nb[1], (float *)last->data, (int64_t)(last->nb[1] / sizeof(float)), M, K, B, (const float *)bias->data, act,
                                              resid ? (const float *)resid->data : nullptr, resid ? (int64_t)(resid->nb[1] / sizeof(float)) : 0, ctx->ws, ctx->ws_size, ctx->stream) == 0) {
            ctx->n_act_shared++; g_act_shared++;
            return used;
        }
        ctx->act_image.key = 0;                                         // (refused before any launch: the plain fused call below quantizes again — ADVICE r5)
    }
    void * ws = ctx->need_ws(need);
    if (!ws) { st = GGML_STATUS_ALLOC_FAILED; return used; }
    act_image_note(ctx, a, M, K, B, b->data, (int64_t)(b->nb[1] / sizeof(float)));
    const int rc = ggml_cdna4_mul_mat_fused((int)a->type, a->data, (int64_t)a->nb[1], (const float *)b->data, (int64_t)(b->nb[1] / sizeof(float)),
                                            (float *)last->data, (int64_t)(last->nb[1] / sizeof(float)), M, K, B, (const float *)bias->data, act,
                                            resid ? (const float *)resid->data : nullptr, resid ? (int64_t)(resid->nb[1] / sizeof(float)) : 0, ws, ctx->ws_size, ctx->stream);
    if (rc) { fprintf(stderr, "ggml-cdna4: fused MUL_MAT failed: %s\n", ggml_cdna4_last_error()); st = GGML_STATUS_FAILED; }
    return used;
}
// ---- one-row MUL_MATs that read the SAME src1 (wq / wk / wv, w_gate / w_up of a decoded token) as ONE launch (round 6).  From the MUL_MAT at node i the
// walk looks a few nodes ahead for plain one-row products of the same activation tensor, weight type and K; a later one may run early (with node i) when nothing between
// the two touches its output's memory — the graph allocator placed that output assuming node order, so it may still hold a tensor that is live until then.  The products that
// rode along are marked done and skipped when the walk reaches them.  Bit-identical to the node-by-node run (ggml_cdna4_mul_mat_group).  GGML_CDNA4_NO_GROUP=1: off.
static std::atomic<int> g_grouped{0};
extern "C" int ggml_backend_cdna4_grouped_count(void) { return g_grouped.load(); }
static bool group_on() { static const bool off = getenv("GGML_CDNA4_NO_GROUP") != nullptr; return !off && !cdna4_exact_mode(); }
static bool plain_one_row_mul_mat(const ggml_tensor * t) {
    if (t->op != GGML_OP_MUL_MAT || ggml_is_empty(t)) return false;
    const ggml_tensor * a = t->src[0], * b = t->src[1];
    if (!is_qweight(a->type) || (a->buffer && cdna4_buft_is_split(a->buffer->buft))) return false;
    if (a->ne[2] != 1 || a->ne[3] != 1 || b->ne[1] != 1 || b->ne[2] != 1 || b->ne[3] != 1 || b->type != GGML_TYPE_F32 || !ggml_is_contiguous(t) || t->type != GGML_TYPE_F32) return false;
    return a->nb[0] == ggml_type_size(a->type) && b->nb[0] == sizeof(float) && !((uintptr_t)b->data & 15);
}
// tries to run node i together with later one-row products of its src1; returns true if node i has been computed (done[] marks every node the launch covered)
static bool try_group_mul_mat(cdna4_backend_ctx * ctx, ggml_cgraph * g, int i, const use_counts & uses, std::vector<char> & done, enum ggml_status & st) {
    ggml_tensor * mm = ggml_graph_node(g, i);
    if (!group_on() || !plain_one_row_mul_mat(mm)) return false;
    const int n_nodes = ggml_graph_n_nodes(g);
    // a member: the product at node k, or the product + the ADD of a row vector behind it (a bias — at one row a residual looks the same and IS the same sum): the launch
    // then writes the ADD's tensor and the product's own tensor is never written (as in try_fused_mul_mat)
    struct member { int k; ggml_tensor * out; const ggml_tensor * bias; };
    auto member_of = [&](int k, member & m) {
        ggml_tensor * t = ggml_graph_node(g, k);
        m.k = k; m.out = t; m.bias = nullptr;
        if (k + 1 < n_nodes && !done[k + 1]) {
            ggml_tensor * n1 = ggml_graph_node(g, k + 1);
            const ggml_tensor * b = n1->op == GGML_OP_ADD && uses.only_reader(t, n1) && ggml_are_same_shape(n1, t) && ggml_is_contiguous(n1) ? other_src(n1, t) : nullptr;
            if (b && is_row_vector_f32(b, t->src[0]->ne[1]) && !mem_overlap(n1, b) && !mem_overlap(n1, t->src[0]) && !mem_overlap(n1, t->src[1])) { m.out = n1; m.bias = b; }
        }
    };
    member ms[4]; int n = 1;
    member_of(i, ms[0]);
    for (int k = i + 1; k < n_nodes && k <= i + 12 && n < 4; k++) {
        ggml_tensor * t = ggml_graph_node(g, k);
        if (done[k] || !plain_one_row_mul_mat(t) || t->src[1] != mm->src[1] || t->src[0]->type != mm->src[0]->type || t->src[0]->ne[0] != mm->src[0]->ne[0]) continue;
        member c; member_of(k, c);
        // the member runs EARLY: what it writes must not overlap anything the nodes in front of it (from i on, outside the group) read or write, the activation row, or
        // another member's output
        bool ok = !mem_overlap(c.out, mm->src[1]);
        for (int q = 0; q < n && ok; q++) if (mem_overlap(c.out, ms[q].out) || (ms[q].bias && mem_overlap(c.out, ms[q].bias))) ok = false;
        for (int q = i; q < k && ok; q++) {
            const ggml_tensor * nq = ggml_graph_node(g, q);
            if (mem_overlap(c.out, nq)) ok = false;
            for (int j = 0; j < GGML_MAX_SRC && ok; j++) if (nq->src[j] && nq->src[j]->data && mem_overlap(c.out, nq->src[j])) ok = false;
        }
        if (ok) ms[n++] = c;
    }
    if (n < 2) return false;
    const void * W[4]; int64_t rb[4], M[4]; float * Y[4]; const float * bias[4];
    for (int q = 0; q < n; q++) {
        const ggml_tensor * t = ggml_graph_node(g, ms[q].k);
        W[q] = t->src[0]->data; rb[q] = (int64_t)t->src[0]->nb[1]; M[q] = t->src[0]->ne[1]; Y[q] = (float *)ms[q].out->data; bias[q] = ms[q].bias ? (const float *)ms[q].bias->data : nullptr;
    }
    const int rc = ggml_cdna4_mul_mat_group((int)mm->src[0]->type, n, W, rb, M, Y, bias, (const float *)mm->src[1]->data, mm->src[0]->ne[0], ctx->stream);
    if (rc == -2) return false;                                         // no grouped form for these: node by node
    if (rc) { fprintf(stderr, "ggml-cdna4: grouped MUL_MAT failed: %s\n", ggml_cdna4_last_error()); st = GGML_STATUS_FAILED; return true; }
    for (int q = 0; q < n; q++) {
        done[ms[q].k] = 1;
        if (ms[q].bias) done[ms[q].k + 1] = 1;
        node_wrote(ctx, ms[q].out);
    }
    ctx->n_grouped += n - 1; g_grouped += n - 1;
    if (ctx->n_grouped == n - 1 && getenv("GGML_CDNA4_STATS")) fprintf(stderr, "ggml-cdna4: %s: %d one-row MUL_MATs of %s run as one launch (first: %s)\n", ctx->name.c_str(), n, mm->src[1]->name, mm->name);
    return true;
}
static ggml_cdna4_tensor tdesc(const ggml_tensor * t) {
    ggml_cdna4_tensor d; d.data = t->data; d.type = (int32_t)t->type; d.reserved = 0;
    for (int k = 0; k < 4; k++) { d.ne[k] = t->ne[k]; d.nb[k] = (int64_t)t->nb[k]; }
    return d;
}
static int try_fused_norm(cdna4_backend_ctx * ctx, ggml_cgraph * g, int i, const use_counts & uses, enum ggml_status & st) {
    ggml_tensor * nm = ggml_graph_node(g, i);
    const int n_nodes = ggml_graph_n_nodes(g);
    if (i + 1 >= n_nodes || !cdna4_ops_supports_tensor(nm)) return 0;
    ggml_tensor * n1 = ggml_graph_node(g, i + 1);
    if (n1->op != GGML_OP_MUL || !uses.only_reader(nm, n1) || !ggml_are_same_shape(n1, nm) || n1->type != GGML_TYPE_F32 || n1->nb[0] != sizeof(float)) return 0;
    const ggml_tensor * gain = other_src(n1, nm);
    if (!is_row_vector_f32(gain, nm->ne[0])) return 0;
    ggml_tensor * last = n1; int used = 2; const ggml_tensor * shift = nullptr;
    if (i + 2 < n_nodes) {
        ggml_tensor * n2 = ggml_graph_node(g, i + 2);
        if (n2->op == GGML_OP_ADD && uses.only_reader(n1, n2) && ggml_are_same_shape(n2, n1) && n2->type == GGML_TYPE_F32 && n2->nb[0] == sizeof(float)) {
            const ggml_tensor * sh = other_src(n2, n1);
            if (is_row_vector_f32(sh, nm->ne[0])) { shift = sh; last = n2; used = 3; }
        }
    }
    if (!alias_or_disjoint(nm->src[0], last) || mem_overlap(last, gain) || (shift && mem_overlap(last, shift))) return 0;
    float eps; memcpy(&eps, nm->op_params, sizeof(float));
    const ggml_cdna4_tensor dx = tdesc(nm->src[0]), dg = tdesc(gain), dd = tdesc(last);
    ggml_cdna4_tensor ds{}; if (shift) ds = tdesc(shift);
    // the node behind the chain is a MUL_MAT that reads its rows and would quantize them to the K-quants' fp16 GEMM image (llama: rms_norm -> mul -> wq | w_gate): the
    // chain's own launch leaves that image in the workspace (ggml_cdna4_op_norm_affine_q8_K: same fp32 rows, same image, bit for bit), so that no MUL_MAT of these rows
    // pays a quantizer launch — the first one takes the hand-off like the others (compute_mul_mat / try_fused_mul_mat)
    if (act_share_on() && i + used < ggml_graph_n_nodes(g)) {
        const ggml_tensor * mmn = ggml_graph_node(g, i + used);
        const ggml_tensor * a = mmn->op == GGML_OP_MUL_MAT ? mmn->src[0] : nullptr;
        // (key 19: the K-quants' Q8_K image, rows of whole 256-value superblocks; 17: the 32-block formats' Q8_0 image — gpt-2's Q4_0 layers at prompt sizes)
        const uint32_t key = a && mmn->src[1] == last && is_qweight(a->type) && last->ne[2] == 1 && last->ne[3] == 1 ? ggml_cdna4_act_image_key((int)a->type, a->ne[1], last->ne[0], last->ne[1]) : 0u;
        if ((key == 19u || key == 17u) && !(a->buffer && cdna4_buft_is_split(a->buffer->buft)) && a->ne[2] == 1 && a->ne[3] == 1 &&
            last->nb[1] == (size_t)last->ne[0] * sizeof(float) && last->ne[0] % (key == 19u ? 256 : 32) == 0 && last->ne[0] <= 8192) {
            const size_t need = ggml_cdna4_mul_mat_workspace_size((int)a->type, last->ne[0], last->ne[1]);
            void * ws = ctx->need_ws(need);
            if (ws && (key == 19u ? ggml_cdna4_op_norm_affine_q8_K : ggml_cdna4_op_norm_affine_q8_0)(&dx, &dg, shift ? &ds : nullptr, &dd, eps, nm->op == GGML_OP_RMS_NORM, (int)a->type, ws, ctx->ws_size, ctx->stream) == 0) {
                act_image_note(ctx, a, a->ne[1], last->ne[0], last->ne[1], last->data, last->ne[0]);
                ctx->act_image.producer = last->data;
                if (ctx->n_act_produced++ == 0 && getenv("GGML_CDNA4_STATS")) fprintf(stderr, "ggml-cdna4: %s: a NORM chain left the activation image of its %lld x %lld rows for %s (key %u)\n", ctx->name.c_str(), (long long)last->ne[1], (long long)last->ne[0], mmn->name, key);
                return used;
            }
        }
    }
    // the fused entry point validates shapes and strides BEFORE its launch: a rejection has written nothing, and the nodes run one by one instead
    // (a layout each separate kernel accepts must not abort a graph that ran before the peephole existed)
    if (ggml_cdna4_op_norm_affine(&dx, &dg, shift ? &ds : nullptr, &dd, eps, nm->op == GGML_OP_RMS_NORM, ctx->stream)) { (void)st; return 0; }
    return used;
}
static int try_fused_soft_max(cdna4_backend_ctx * ctx, ggml_cgraph * g, int i, const use_counts & uses, enum ggml_status & st) {
    ggml_tensor * sc = ggml_graph_node(g, i);
    if (i + 2 >= ggml_graph_n_nodes(g) || !cdna4_ops_supports_tensor(sc)) return 0;
    ggml_tensor * dm = ggml_graph_node(g, i + 1), * sm = ggml_graph_node(g, i + 2);
    if (dm->op != GGML_OP_DIAG_MASK_INF || sm->op != GGML_OP_SOFT_MAX || dm->src[0] != sc || sm->src[0] != dm || sm->src[1]) return 0;
    if (!uses.only_reader(sc, dm) || !uses.only_reader(dm, sm) || !cdna4_ops_supports_tensor(dm) || !cdna4_ops_supports_tensor(sm)) return 0;
    float pre, scale, max_bias;
    memcpy(&pre, sc->op_params, 4); memcpy(&scale, (const float *)sm->op_params + 0, 4); memcpy(&max_bias, (const float *)sm->op_params + 1, 4);
    const int n_past = ((const int32_t *)dm->op_params)[0];
    if (n_past < 0 || !alias_or_disjoint(sc->src[0], sm)) return 0;
    const ggml_cdna4_tensor dx = tdesc(sc->src[0]), dd = tdesc(sm);
    if (ggml_cdna4_op_soft_max_ext(&dx, nullptr, &dd, scale, max_bias, 1, pre, n_past, ctx->stream)) { (void)st; return 0; }      // (rejected before any launch: node by node, as above)
    return 3;
}

// everything a node's launches depend on: op, types, shapes, strides, addresses, parameters — of the node and of its sources
static uint64_t graph_signature(ggml_cgraph * g) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    auto mix = [&](uint64_t w) { h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; };
    auto words = [&](const void * p, size_t nbytes) { uint64_t w; for (size_t i = 0; i + 8 <= nbytes; i += 8) { memcpy(&w, (const char *)p + i, 8); mix(w); } };
    auto tensor = [&](const ggml_tensor * t) {
        mix((uint64_t)t->type); words(t->ne, sizeof(t->ne)); words(t->nb, sizeof(t->nb)); mix((uint64_t)(uintptr_t)t->data);
    };
    const int n = ggml_graph_n_nodes(g);
    mix((uint64_t)n);
    for (int i = 0; i < n; i++) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        mix(((uint64_t)t->op << 32) | (uint32_t)t->flags); words(t->op_params, sizeof(t->op_params)); tensor(t);
        for (int j = 0; j < GGML_MAX_SRC; j++) { const ggml_tensor * s = t->src[j]; if (!s) continue; mix((uint64_t)(uintptr_t)s + j); tensor(s); }
    }
    return h ? h : 1;
}
static bool graph_has_split_weights(ggml_cgraph * g) {
    for (int i = 0; i < ggml_graph_n_nodes(g); i++) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        if (t->op == GGML_OP_MUL_MAT && t->src[0]->buffer && cdna4_buft_is_split(t->src[0]->buffer->buft)) return true;
    }
    return false;
}

static enum ggml_status run_nodes(cdna4_backend_ctx * ctx, ggml_cgraph * cgraph) {
    const int n_nodes = ggml_graph_n_nodes(cgraph);
    static const bool no_fuse = getenv("GGML_CDNA4_NO_FUSE") != nullptr || cdna4_exact_mode();      // (exact mode runs node by node: its kernels have no fused forms)
    const bool fuse = !no_fuse && n_nodes > 1;
    const use_counts uses = fuse ? use_counts(cgraph) : use_counts(nullptr, 0);
    ctx->act_image.key = 0;                                              // (activations of an earlier graph: the host may have rewritten them since)
    ctx->moe_front.key = 0;
    std::vector<char> done(fuse ? n_nodes : 0, 0);                      // one-row MUL_MATs that already ran in an earlier node's launch (try_group_mul_mat)
    for (int i = 0; i < n_nodes; i++) {
        ggml_tensor * node = ggml_graph_node(cgraph, i);
        if (ggml_is_empty(node)) continue;
        if (fuse && done[i]) continue;
        enum ggml_status st = GGML_STATUS_SUCCESS;
        if (fuse) {
            if (node->op == GGML_OP_MUL_MAT && try_group_mul_mat(ctx, cgraph, i, uses, done, st)) {
                if (st != GGML_STATUS_SUCCESS) return st;
                continue;                                               // (done[] covers the group's nodes, this one's bias ADD included)
            }
            int used = 0;
            if (node->op == GGML_OP_MUL_MAT) used = try_fused_mul_mat(ctx, cgraph, i, uses, st);
            else if (node->op == GGML_OP_NORM || node->op == GGML_OP_RMS_NORM) used = try_fused_norm(ctx, cgraph, i, uses, st);
            else if (node->op == GGML_OP_SCALE) used = try_fused_soft_max(ctx, cgraph, i, uses, st);
            if (st != GGML_STATUS_SUCCESS) return st;
            if (used) {
                ggml_tensor * last = ggml_graph_node(cgraph, i + used - 1);     // (a chain writes its last node only)
                if (ctx->act_image.producer == last->data) ctx->act_image.producer = nullptr;      // (the chain itself made the image of what it wrote)
                else node_wrote(ctx, last);
                i += used - 1; continue;
            }
        }
        switch (node->op) {
            case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: break;
            case GGML_OP_MUL_MAT:    st = compute_mul_mat(ctx, node); break;
            case GGML_OP_MUL_MAT_ID: st = compute_mul_mat_id(ctx, node); break;
            default:                 st = cdna4_ops_compute(ctx, node); break;
        }
        if (st != GGML_STATUS_SUCCESS) {
            fprintf(stderr, "ggml-cdna4: op %s (%s) failed\n", ggml_op_name(node->op), node->name);
            return st;
        }
        if (node->op != GGML_OP_NONE && node->op != GGML_OP_RESHAPE && node->op != GGML_OP_VIEW && node->op != GGML_OP_PERMUTE && node->op != GGML_OP_TRANSPOSE)
            node_wrote(ctx, node);
    }
    return GGML_STATUS_SUCCESS;
}

// A graph that comes back UNCHANGED (same nodes, shapes, addresses and parameters: graph_signature) is captured into a HIP graph on its
// second appearance and replayed from then on — one hipGraphLaunch instead of one launch per node (the counterpart of the CUDA-graph
// path of ggml-cuda.cu:2417-2694, without its parameter patching: a graph that changes, like a decode step whose KV views move every
// token, simply stays on plain launches).  CDNA4_GRAPH_SLOTS graphs are remembered at a time, the least recently used one replaced,
// so that the splits ggml_backend_sched cuts one model graph into (each its own graph_compute call, in turn) all replay.  The first
// appearance always runs eagerly, which also sizes every workspace, so that the capture itself allocates nothing.  Everything the
// kernels need is capturable: one stream, no host synchronisation, and the split-K exchange flags are reset by their readers
// (gemm_w8_epilogue.inc / gemm_kq_t64.inc).  Row-split weights (several streams and devices) stay on plain launches.
// GGML_CDNA4_NO_GRAPHS=1 turns it off; a failed capture turns it off for the backend instance.
static enum ggml_status cdna4_backend_graph_compute(ggml_backend_t backend, ggml_cgraph * cgraph) {
    cdna4_backend_ctx * ctx = (cdna4_backend_ctx *)backend->context;
    HIP_OK(hipSetDevice(ctx->device));
    // an earlier launch on a route that waits for co-resident work-groups (GGML_CDNA4_OWNED_DEVICE=1 only) gave up waiting: what it wrote holds NaN tiles.  Said HERE, as a
    // status — never a silently wrong tensor — and the kernel library has switched itself to the non-waiting routes (captured graphs hold the old routes: dropped)
    if (const int fault = ggml_cdna4_device_fault(1)) {
        fprintf(stderr, "ggml-cdna4: %s: an earlier graph waited for a co-resident work-group that never arrived (code %d) — the device is shared; its results are invalid. "
                        "Continuing on the non-waiting routes.\n", ctx->name.c_str(), fault);
        for (auto & gs : ctx->graph_slots) { if (gs.exec) (void)hipGraphExecDestroy(gs.exec); gs.exec = nullptr; gs.sig = 0; }
        return GGML_STATUS_FAILED;
    }
    static const bool no_graphs = getenv("GGML_CDNA4_NO_GRAPHS") != nullptr || cdna4_exact_mode();      // (the parity mode runs node by node, as ggml_cdna4_ops.h and exact.hip say: ADVICE r4)
    if (no_graphs || ctx->graphs_off || ggml_graph_n_nodes(cgraph) < 2) return run_nodes(ctx, cgraph);
    uint64_t sig = graph_signature(cgraph);
    if (sig == 0) sig = 1;                                                     // (0 marks an unused slot)
    // addresses and decisions the captured launches hold beyond the tensors': this context's workspace, the kernel library's scratch, and whether a weight's resident
    // image was found and where (ggml_cdna4_scratch_generation also moves with every image registered or unregistered: ADVICE r5 — a partial set_tensor that drops an
    // image, or a freed Resident buffer whose successor lands at the same address, must not replay launches that read the old image)
    const uint64_t gen = ggml_cdna4_scratch_generation() * 0x9E3779B97F4A7C15ull + ctx->ws_gen;
    cdna4_backend_ctx::graph_slot * slot = nullptr, * lru = &ctx->graph_slots[0];
    for (auto & gs : ctx->graph_slots) {
        if (gs.sig == sig) { slot = &gs; break; }
        if (gs.last_use < lru->last_use) lru = &gs;
    }
    const uint64_t now = ++ctx->graph_clock;
    if (slot && slot->exec && slot->gen == gen) {
        slot->last_use = now;
        HIP_OK(hipGraphLaunch(slot->exec, ctx->stream));
        ctx->n_graph_launches++;
        return GGML_STATUS_SUCCESS;
    }
    if (!slot) {                                                               // first appearance: remember it, run it eagerly
        if (lru->exec) { (void)hipGraphExecDestroy(lru->exec); lru->exec = nullptr; }
        lru->sig = sig; lru->last_use = now;
        return run_nodes(ctx, cgraph);
    }
    slot->last_use = now;
    // captured against a workspace or scratch block that has moved since (both only ever grow, and this graph has run before: the
    // new capture allocates nothing)
    if (slot->exec) { (void)hipGraphExecDestroy(slot->exec); slot->exec = nullptr; }
    if (graph_has_split_weights(cgraph)) return run_nodes(ctx, cgraph);
    // second appearance: capture, instantiate, launch
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); ctx->graphs_off = true; return run_nodes(ctx, cgraph); }
    const enum ggml_status st = run_nodes(ctx, cgraph);
    const hipError_t ee = hipStreamEndCapture(ctx->stream, &graph);
    if (st != GGML_STATUS_SUCCESS || ee != hipSuccess || !graph || hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        slot->exec = nullptr; ctx->graphs_off = true;
        fprintf(stderr, "ggml-cdna4: HIP-graph capture failed, staying on plain launches\n");
        return run_nodes(ctx, cgraph);                                     // nothing has run yet: the capture only recorded
    }
    (void)hipGraphDestroy(graph);
    ctx->n_graph_captures++;
    slot->gen = ggml_cdna4_scratch_generation() * 0x9E3779B97F4A7C15ull + ctx->ws_gen;     // (the eager first run sized everything: unchanged by the capture)
    HIP_OK(hipGraphLaunch(slot->exec, ctx->stream));
    ctx->n_graph_launches++;
    return GGML_STATUS_SUCCESS;
}
// ---- asynchronous tensor access and events: what ggml_backend_sched uses to overlap a split's input copies with the previous
//      split's compute (src/ggml-backend.cpp:1361-1443); counterparts: ggml-cuda.cu:2353-2406, 2772-2809
static bool backend_is_cdna4(ggml_backend_t backend);
static void cdna4_backend_set_tensor_async(ggml_backend_t backend, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    cdna4_backend_ctx * ctx = (cdna4_backend_ctx *)backend->context;
    ggml_backend_buffer_t buf = tensor->view_src ? tensor->view_src->buffer : tensor->buffer;
    GGML_ASSERT(buffer_is_cdna4(buf) && "set_tensor_async: the tensor must live in a buffer of this backend");
    HIP_OK(hipSetDevice(ctx->device));
    cdna4_buffer_ctx * bctx = (cdna4_buffer_ctx *)buf->context;
    // (buffer_from_host_ptr: the destination is the HOST's own memory — a plain copy behind whatever the stream still reads from it)
    if (bctx->host_registered) { HIP_OK(hipStreamSynchronize(ctx->stream)); memcpy((char *)tensor->data + offset, data, size); return; }
    HIP_OK(hipMemcpyAsync((char *)tensor->data + offset, data, size, hipMemcpyHostToDevice, ctx->stream));
    if (bctx->resident && resident_find(bctx, (const char *)tensor->data + offset, size)) { HIP_OK(hipStreamSynchronize(ctx->stream)); resident_written(bctx, tensor, offset, size); }   // (a weight with an image: the image follows its bytes)
}
static void cdna4_backend_get_tensor_async(ggml_backend_t backend, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    cdna4_backend_ctx * ctx = (cdna4_backend_ctx *)backend->context;
    ggml_backend_buffer_t buf = tensor->view_src ? tensor->view_src->buffer : tensor->buffer;
    GGML_ASSERT(buffer_is_cdna4(buf) && "get_tensor_async: the tensor must live in a buffer of this backend");
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipMemcpyAsync(data, (const char *)tensor->data + offset, size, hipMemcpyDeviceToHost, ctx->stream));
}
static bool cdna4_backend_cpy_tensor_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, const ggml_tensor * src, ggml_tensor * dst) {
    if (!backend_is_cdna4(backend_src) || !backend_is_cdna4(backend_dst)) return false;
    ggml_backend_buffer_t sbuf = src->view_src ? src->view_src->buffer : src->buffer, dbuf = dst->view_src ? dst->view_src->buffer : dst->buffer;
    if (!buffer_is_cdna4(sbuf) || !buffer_is_cdna4(dbuf)) return false;
    cdna4_backend_ctx * sctx = (cdna4_backend_ctx *)backend_src->context, * dctx = (cdna4_backend_ctx *)backend_dst->context;
    const int sdev = ((cdna4_buffer_ctx *)sbuf->context)->device, ddev = ((cdna4_buffer_ctx *)dbuf->context)->device;
    if (sdev != sctx->device || ddev != dctx->device) return false;
    if (backend_src == backend_dst) {
        HIP_OK(hipSetDevice(sctx->device));
        HIP_OK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(dst), hipMemcpyDeviceToDevice, sctx->stream));
        return true;
    }
    // the copy runs on the SOURCE stream (behind whatever produced src); the destination stream then waits for it
    HIP_OK(hipSetDevice(sctx->device));
    if (sdev == ddev) HIP_OK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(dst), hipMemcpyDeviceToDevice, sctx->stream));
    else HIP_OK(hipMemcpyPeerAsync(dst->data, ddev, src->data, sdev, ggml_nbytes(dst), sctx->stream));
    if (!sctx->ev_copy) HIP_OK(hipEventCreateWithFlags(&sctx->ev_copy, hipEventDisableTiming));
    HIP_OK(hipEventRecord(sctx->ev_copy, sctx->stream));
    HIP_OK(hipSetDevice(dctx->device));
    HIP_OK(hipStreamWaitEvent(dctx->stream, sctx->ev_copy, 0));
    return true;
}
static void cdna4_backend_event_record(ggml_backend_t backend, ggml_backend_event_t event) {
    cdna4_backend_ctx * ctx = (cdna4_backend_ctx *)backend->context;
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipEventRecord((hipEvent_t)event->context, ctx->stream));
}
static void cdna4_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event) {
    cdna4_backend_ctx * ctx = (cdna4_backend_ctx *)backend->context;
    if (!backend_is_cdna4(backend)) GGML_ABORT("event_wait on a foreign backend");
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)event->context, 0));
}
static const ggml_backend_i cdna4_backend_iface = {
    /* .get_name           = */ cdna4_backend_get_name,
    /* .free               = */ cdna4_backend_free,
    /* .set_tensor_async   = */ cdna4_backend_set_tensor_async,
    /* .get_tensor_async   = */ cdna4_backend_get_tensor_async,
    /* .cpy_tensor_async   = */ cdna4_backend_cpy_tensor_async,
    /* .synchronize        = */ cdna4_backend_synchronize,
    /* .graph_plan_create  = */ NULL,
    /* .graph_plan_free    = */ NULL,
    /* .graph_plan_update  = */ NULL,
    /* .graph_plan_compute = */ NULL,
    /* .graph_compute      = */ cdna4_backend_graph_compute,
    /* .event_record       = */ cdna4_backend_event_record,
    /* .event_wait         = */ cdna4_backend_event_wait,
};
static bool backend_is_cdna4(ggml_backend_t backend) { return backend && backend->iface.get_name == cdna4_backend_get_name; }
static ggml_guid_t cdna4_guid(void) {
    static ggml_guid guid = {0x63, 0x64, 0x6e, 0x61, 0x34, 0x2d, 0x6d, 0x69, 0x33, 0x35, 0x35, 0x78, 0x2d, 0x67, 0x67, 0x01};
    return &guid;
}

// used by the op table to get the stream / scratch without knowing the context layout
extern "C" void * cdna4_backend_stream(void * ctx) { return (void *)((cdna4_backend_ctx *)ctx)->stream; }
extern "C" void * cdna4_backend_scratch(void * ctx, size_t n) { return ((cdna4_backend_ctx *)ctx)->need_ws(n); }

// ============================================================================================================
// device
static const char * cdna4_dev_get_name(ggml_backend_dev_t dev) { return ((cdna4_device_ctx *)dev->context)->name.c_str(); }
static bool cdna4_host_ptr_buffers_on() { static const bool on = getenv("GGML_CDNA4_HOST_PTR_BUFFERS") && atoi(getenv("GGML_CDNA4_HOST_PTR_BUFFERS")) != 0; return on; }
static const char * cdna4_dev_get_description(ggml_backend_dev_t dev) { return ((cdna4_device_ctx *)dev->context)->description.c_str(); }
static void cdna4_dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    HIP_OK(hipSetDevice(((cdna4_device_ctx *)dev->context)->device));
    HIP_OK(hipMemGetInfo(free, total));
}
static enum ggml_backend_dev_type cdna4_dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static void cdna4_dev_get_props(ggml_backend_dev_t dev, ggml_backend_dev_props * props) {
    props->name = cdna4_dev_get_name(dev);
    props->description = cdna4_dev_get_description(dev);
    props->type = GGML_BACKEND_DEVICE_TYPE_GPU;
    cdna4_dev_get_memory(dev, &props->memory_free, &props->memory_total);
    // buffer_from_host_ptr stays FALSE unless asked for (GGML_CDNA4_HOST_PTR_BUFFERS=1), like the reference's GPU backends (ggml-cuda.cu:2919 leaves the cap false and the
    // slot NULL): llama.cpp checks the cap when use_mmap is on (its default) and would wrap the whole weight file — every offloaded weight then read over PCIe (63 GB/s against
    // 8 TB/s), or, where hipHostRegister refuses the PROT_READ file mapping, "unable to allocate buffer" instead of a device buffer + copies
    props->caps = { /* async */ true, /* host_buffer */ true, /* buffer_from_host_ptr */ cdna4_host_ptr_buffers_on(), /* events */ true };
}
static ggml_backend_t cdna4_dev_init_backend(ggml_backend_dev_t dev, const char *) {
    cdna4_device_ctx * dctx = (cdna4_device_ctx *)dev->context;
    if (hipSetDevice(dctx->device) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    cdna4_backend_ctx * ctx = new cdna4_backend_ctx;
    ctx->device = dctx->device; ctx->name = dctx->name;
    if (hipStreamCreate(&ctx->stream) != hipSuccess) { (void)hipGetLastError(); delete ctx; return NULL; }
    return new ggml_backend{ /* .guid = */ cdna4_guid(), /* .iface = */ cdna4_backend_iface, /* .device = */ dev, /* .context = */ ctx };
}
static ggml_backend_buffer_type_t cdna4_dev_get_buffer_type(ggml_backend_dev_t dev) { return cdna4_buffer_type(((cdna4_device_ctx *)dev->context)->device); }

static bool cdna4_dev_supports_op(ggml_backend_dev_t, const ggml_tensor * op) {
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: return true;
        case GGML_OP_MUL_MAT: return supports_mul_mat(op);
        case GGML_OP_MUL_MAT_ID: return supports_mul_mat_id(op);
        default: return cdna4_ops_supports_tensor(op);
    }
}
static bool cdna4_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    if (cdna4_buft_is_split(buft)) return buft->device == dev;          // row-split weights: consumed by the backend of their main device
    return buft_is_cdna4(buft) && ((cdna4_buft_ctx *)buft->context)->device == ((cdna4_device_ctx *)dev->context)->device;
}
static bool cdna4_dev_offload_op(ggml_backend_dev_t, const ggml_tensor * op) {
    // worth moving host-resident weights over PCIe only for batched work (cf. ggml-cuda.cu:3241-3247)
    const int64_t batch = op->op == GGML_OP_MUL_MAT_ID ? op->ne[2] : op->ne[1];
    return (op->op == GGML_OP_MUL_MAT || op->op == GGML_OP_MUL_MAT_ID) && batch >= 32;
}
// ---- pinned host buffer type: what the scheduler / applications put CPU-side tensors in so that copies to the device are true
//      DMA transfers (counterpart: ggml_backend_cuda_host_buffer_type, ggml-cuda.cu:1041-1118).  The buffer itself is ggml's
//      CPU buffer over hipHostMalloc'ed memory.
static const char * cdna4_host_buft_get_name(ggml_backend_buffer_type_t) { return "CDNA4_Host"; }
static void cdna4_host_buffer_free(ggml_backend_buffer_t buffer) { HIP_OK(hipHostFree(buffer->context)); }
static ggml_backend_buffer_t cdna4_host_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    void * ptr = nullptr;
    if (hipHostMalloc(&ptr, size > 0 ? size : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return ggml_backend_buft_alloc_buffer(ggml_backend_cpu_buffer_type(), size);      // pageable memory still works, just slower
    }
    ggml_backend_buffer_t buffer = ggml_backend_cpu_buffer_from_ptr(ptr, size);
    buffer->buft = buft;
    buffer->iface.free_buffer = cdna4_host_buffer_free;
    return buffer;
}
static size_t cdna4_host_buft_get_alignment(ggml_backend_buffer_type_t) { return ggml_backend_buft_get_alignment(ggml_backend_cpu_buffer_type()); }
static bool cdna4_host_buft_is_host(ggml_backend_buffer_type_t) { return true; }
static ggml_backend_buffer_type_t cdna4_dev_get_host_buffer_type(ggml_backend_dev_t dev) {
    static ggml_backend_buffer_type host_buft = {
        /* .iface   = */ { cdna4_host_buft_get_name, cdna4_host_buft_alloc_buffer, cdna4_host_buft_get_alignment, NULL, NULL, cdna4_host_buft_is_host },
        /* .device  = */ nullptr,
        /* .context = */ nullptr,
    };
    if (!host_buft.device) host_buft.device = dev;
    return &host_buft;
}
// ---- buffer_from_host_ptr (src/ggml-backend-impl.h:163): a buffer over memory the HOST owns — e.g. the mapping of a model file — registered for device access
// (hipHostRegister: the kernels read it in place over the host link; no copy, no second footprint).  ggml places tensors in such a buffer by HOST address, so the device
// must see the range at the same address: where hipHostGetDevicePointer answers otherwise the request is declined (NULL: the caller falls back to a device buffer + copies).
// Meant for weights that are read rarely or must not be duplicated; prefill over the host link is bound by it (PCIe Gen5 x16: 63 GB/s).
static ggml_backend_buffer_t cdna4_dev_buffer_from_host_ptr(ggml_backend_dev_t dev, void * ptr, size_t size, size_t /* max_tensor_size */) {
    const int device = ((cdna4_device_ctx *)dev->context)->device;
    if (!cdna4_host_ptr_buffers_on()) return NULL;                      // (opt-in: see get_props)
    if (!ptr || size == 0 || hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    if (hipHostRegister(ptr, size, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    void * dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, ptr, 0) != hipSuccess || dptr != ptr) { (void)hipGetLastError(); (void)hipHostUnregister(ptr); return NULL; }
    cdna4_buffer_ctx * ctx = new cdna4_buffer_ctx{device, ptr, size};
    ctx->host_registered = true;
    return ggml_backend_buffer_init(cdna4_buffer_type(device), cdna4_buffer_iface, ctx, size);
}
static ggml_backend_event_t cdna4_dev_event_new(ggml_backend_dev_t dev) {
    HIP_OK(hipSetDevice(((cdna4_device_ctx *)dev->context)->device));
    hipEvent_t ev = nullptr;
    HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return new ggml_backend_event{ /* .device = */ dev, /* .context = */ ev };
}
static void cdna4_dev_event_free(ggml_backend_dev_t, ggml_backend_event_t event) { (void)hipEventDestroy((hipEvent_t)event->context); delete event; }
static void cdna4_dev_event_synchronize(ggml_backend_dev_t, ggml_backend_event_t event) { HIP_OK(hipEventSynchronize((hipEvent_t)event->context)); }

static const ggml_backend_device_i cdna4_device_iface = {
    /* .get_name             = */ cdna4_dev_get_name,
    /* .get_description      = */ cdna4_dev_get_description,
    /* .get_memory           = */ cdna4_dev_get_memory,
    /* .get_type             = */ cdna4_dev_get_type,
    /* .get_props            = */ cdna4_dev_get_props,
    /* .init_backend         = */ cdna4_dev_init_backend,
    /* .get_buffer_type      = */ cdna4_dev_get_buffer_type,
    /* .get_host_buffer_type = */ cdna4_dev_get_host_buffer_type,
    /* .buffer_from_host_ptr = */ cdna4_dev_buffer_from_host_ptr,
    /* .supports_op          = */ cdna4_dev_supports_op,
    /* .supports_buft        = */ cdna4_dev_supports_buft,
    /* .offload_op           = */ cdna4_dev_offload_op,
    /* .event_new            = */ cdna4_dev_event_new,
    /* .event_free           = */ cdna4_dev_event_free,
    /* .event_synchronize    = */ cdna4_dev_event_synchronize,
};

// ============================================================================================================
// registry
struct cdna4_reg_ctx {
    std::vector<ggml_backend_device> devices;
    std::vector<cdna4_device_ctx> dctx;
    ggml_backend_buffer_type bufts[CDNA4_MAX_DEVICES];
    cdna4_buft_ctx buft_ctx[CDNA4_MAX_DEVICES];
    ggml_backend_buffer_type res_bufts[CDNA4_MAX_DEVICES];              // CDNA4_Resident<i>: weights + their kernel-native images
    cdna4_buft_ctx res_buft_ctx[CDNA4_MAX_DEVICES];
    ggml_backend_buffer_type_t extra[CDNA4_MAX_DEVICES][2];             // what ggml_backend_dev_get_extra_bufts returns per device (NULL-terminated)
    int n = 0;
};
static cdna4_reg_ctx * g_reg = nullptr;

static ggml_backend_buffer_type_t cdna4_buffer_type(int device) {
    GGML_ASSERT(g_reg && device >= 0 && device < g_reg->n);
    return &g_reg->bufts[device];
}
int cdna4_reg_device_count(void) { (void)ggml_backend_cdna4_reg(); return g_reg ? g_reg->n : 0; }
ggml_backend_dev_t cdna4_reg_device(int i) { (void)ggml_backend_cdna4_reg(); GGML_ASSERT(g_reg && i >= 0 && i < g_reg->n); return &g_reg->devices[i]; }
static const char * cdna4_reg_get_name(ggml_backend_reg_t) { return "CDNA4"; }
static size_t cdna4_reg_get_device_count(ggml_backend_reg_t reg) { return (size_t)((cdna4_reg_ctx *)reg->context)->n; }
static ggml_backend_dev_t cdna4_reg_get_device(ggml_backend_reg_t reg, size_t index) {
    cdna4_reg_ctx * ctx = (cdna4_reg_ctx *)reg->context;
    GGML_ASSERT(index < (size_t)ctx->n);
    return &ctx->devices[index];
}
static ggml_backend_feature g_features[] = { {"WAVE64", "1"}, {"MFMA_F16", "1"}, {"INT8_DOT", "1"}, {nullptr, nullptr} };
static ggml_backend_feature * cdna4_get_features(ggml_backend_reg_t) { return g_features; }
// the device's extra buffer types (ggml_backend_dev_get_extra_bufts_t, include/ggml-backend.h:192; the CPU backend's: src/ggml-cpu/ggml-cpu.cpp:581-582): hosts try them
// for weights first.  Ours is CDNA4_Resident<i>: an ordinary device buffer whose re-encoded formats (Q5_0 / IQ4_NL / Q4_1 / Q5_1 / Q3_K / Q2_K / IQ4_XS) also keep their
// kernel-native image, built once at set_tensor — prefill launches no conversion any more; everything else about the buffer is the default type's.
static ggml_backend_buffer_type_t * cdna4_dev_get_extra_bufts(ggml_backend_dev_t dev) { return g_reg->extra[((cdna4_device_ctx *)dev->context)->device]; }
static ggml_backend_buffer_type_t cdna4_resident_buffer_type(int device) { (void)ggml_backend_cdna4_reg(); return (g_reg && device >= 0 && device < g_reg->n) ? &g_reg->res_bufts[device] : nullptr; }
static void * cdna4_reg_get_proc_address(ggml_backend_reg_t, const char * name) {
    if (strcmp(name, "ggml_backend_dev_get_extra_bufts") == 0) return (void *)cdna4_dev_get_extra_bufts;
    if (strcmp(name, "ggml_backend_cdna4_resident_buffer_type") == 0) return (void *)cdna4_resident_buffer_type;     // (int device) -> the buffer type directly
    if (strcmp(name, "ggml_backend_get_features") == 0) return (void *)cdna4_get_features;
    if (strcmp(name, "ggml_backend_cdna4_moe_front_shared_count") == 0) return (void *)ggml_backend_cdna4_moe_front_shared_count;   // MUL_MAT_IDs that multiplied the previous one's front (statistics)
    if (strcmp(name, "ggml_backend_cdna4_grouped_count") == 0) return (void *)ggml_backend_cdna4_grouped_count;       // one-row MUL_MATs that rode in another product's launch (statistics)
    if (strcmp(name, "ggml_backend_cdna4_act_shared_count") == 0) return (void *)ggml_backend_cdna4_act_shared_count;   // MUL_MATs that reused the previous one's quantized activations (statistics)
    if (strcmp(name, "ggml_backend_cdna4_ksplit_buffer_type") == 0) return (void *)cdna4_ksplit_buffer_type;   // the K-split counterpart (no reference equivalent; ggml_cdna4_split.cpp)
    if (strcmp(name, "ggml_backend_split_buffer_type") == 0) return (void *)cdna4_split_buffer_type;      // ggml_backend_split_buffer_type_t, include/ggml-backend.h:188
    if (strcmp(name, "ggml_backend_cdna4_split_ranges") == 0) return (void *)ggml_backend_cdna4_split_ranges;   // the partition of either split, as pure arithmetic (ggml_cdna4_split.cpp)
    return NULL;
}
static const ggml_backend_reg_i cdna4_reg_iface = {
    /* .get_name         = */ cdna4_reg_get_name,
    /* .get_device_count = */ cdna4_reg_get_device_count,
    /* .get_device       = */ cdna4_reg_get_device,
    /* .get_proc_address = */ cdna4_reg_get_proc_address,
};

static ggml_backend_reg_t ggml_backend_cdna4_reg(void) {
    static ggml_backend_reg reg;
    static std::once_flag once;
    std::call_once(once, [] {
        cdna4_reg_ctx * ctx = new cdna4_reg_ctx;
        int n = ggml_cdna4_device_count();
        if (n > CDNA4_MAX_DEVICES) n = CDNA4_MAX_DEVICES;
        ctx->n = n;
        ctx->dctx.resize(n); ctx->devices.resize(n);
        reg = ggml_backend_reg{ /* .api_version = */ GGML_BACKEND_API_VERSION, /* .iface = */ cdna4_reg_iface, /* .context = */ ctx };
        for (int i = 0; i < n; i++) {
            hipDeviceProp_t prop;
            std::string desc = "AMD GPU";
            if (hipGetDeviceProperties(&prop, i) == hipSuccess) desc = std::string(prop.name) + " (" + prop.gcnArchName + ")";
            ctx->dctx[i] = cdna4_device_ctx{i, "CDNA4" + std::to_string(i), desc};
            ctx->devices[i] = ggml_backend_device{ /* .iface = */ cdna4_device_iface, /* .reg = */ &reg, /* .context = */ &ctx->dctx[i] };
            ctx->buft_ctx[i] = cdna4_buft_ctx{i, "CDNA4" + std::to_string(i)};
            ctx->bufts[i] = ggml_backend_buffer_type{ /* .iface = */ cdna4_buft_iface, /* .device = */ &ctx->devices[i], /* .context = */ &ctx->buft_ctx[i] };
            ctx->res_buft_ctx[i] = cdna4_buft_ctx{i, "CDNA4_Resident" + std::to_string(i), true};
            ctx->res_bufts[i] = ggml_backend_buffer_type{ /* .iface = */ cdna4_buft_iface, /* .device = */ &ctx->devices[i], /* .context = */ &ctx->res_buft_ctx[i] };
            ctx->extra[i][0] = &ctx->res_bufts[i]; ctx->extra[i][1] = nullptr;
        }
        g_reg = ctx;
    });
    return &reg;
}

extern "C" {
GGML_BACKEND_API ggml_backend_reg_t ggml_backend_init(void) { return ggml_backend_cdna4_reg(); }
GGML_BACKEND_API int ggml_backend_score(void) { return ggml_cdna4_device_count() > 0 ? 100 : 0; }
// static-registration entry point for an in-tree build (`ggml_backend_register(ggml_backend_cdna4_reg_public())`)
GGML_BACKEND_API ggml_backend_reg_t ggml_backend_cdna4_reg_public(void) { return ggml_backend_cdna4_reg(); }
}
