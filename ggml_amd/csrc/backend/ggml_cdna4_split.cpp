// ggml_cdna4_split.cpp — the row-split buffer type of the MI355X plug-in and the multi-device MUL_MAT that consumes it.
//
// What it replaces: ggml_backend_cuda_split_buffer_type and the `split` path of ggml_cuda_op_mul_mat
// (/root/reference/src/ggml-cuda/ggml-cuda.cu:714-1040, 1353-1666; typedef include/ggml-backend.h:188).  Semantics kept: a 2-D
// weight tensor placed in this buffer type has its ROWS (= output features) partitioned into contiguous ranges, one per GPU;
// it must be written and read whole (set_tensor / get_tensor scatter / gather the ranges); views of it are not supported;
// only MUL_MAT may consume it, with the activations and the result living on the main device.
// What is different (MI355X-first): shard boundaries are multiples of 128 rows (the GEMM tile), every shard runs the SAME
// C-ABI call the single-GPU path runs (ggml_cdna4_mul_mat: activation quantizer + GEMV / MFMA GEMM) on its own stream of its
// own device — no chunking of the activation columns, no per-chunk events: one peer copy of X in, one strided peer copy of
// the [B][rows] result out per shard, over xGMI; the main device's shard writes straight into dst.
// GGML_CDNA4_SPLIT_SELF=N (tests on a one-GPU box): N shards, all on the main device — every code path of the scatter / gather /
// lane machinery runs, the peer copies become local ones.
//
// K-SPLIT (round 4; proc address "ggml_backend_cdna4_ksplit_buffer_type", same signature): the north star's "all-reduce on the activations".  The
// reduction dimension K is partitioned instead of the rows: shard i holds columns [klo_i, khi_i) of EVERY row (whole 256-weight superblocks: a byte
// range of each block-quantized row, scattered by set_tensor with one 2-D copy per shard), multiplies them with its slice of the activations — no
// copy at all on the main device: the kernel library takes a row stride — and produces a full-size partial [B][M]; the partials are summed by ONE
// RCCL all-reduce over xGMI (librccl loaded at run time; the main device receives straight into dst), or, where the shards share a device
// (GGML_CDNA4_SPLIT_SELF) or RCCL is absent, by peer copies + ggml_cdna4_sum_partials in shard order.  A row-split layer followed by a K-split layer
// needs no collective in between (the first one's output shard IS the second one's activation slice); this type is the second half of such a pair.
#include "ggml_cdna4_internal.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

struct split_buft_ctx {
    int main_device; int n_shards; bool self; bool ksplit;
    float bound[CDNA4_MAX_DEVICES + 1];            // cumulative row fractions, bound[0] = 0, bound[n] = 1
    int shard_dev[CDNA4_MAX_DEVICES];
    std::string name;
};
struct split_tensor {                              // tensor->extra; lo / hi: row range (row split) or K range (K split) of the shard
    int n = 0; bool ksplit = false;
    int dev[CDNA4_MAX_DEVICES]; void * data[CDNA4_MAX_DEVICES]; int64_t lo[CDNA4_MAX_DEVICES], hi[CDNA4_MAX_DEVICES];
};
struct split_buffer_ctx { std::vector<split_tensor *> tensors; };

static int self_shards() {
    static const int n = getenv("GGML_CDNA4_SPLIT_SELF") ? atoi(getenv("GGML_CDNA4_SPLIT_SELF")) : 0;
    return n > CDNA4_MAX_DEVICES ? CDNA4_MAX_DEVICES : n;
}
// rows [lo, hi) of shard i: fractions of the row count, rounded DOWN to the 128-row GEMM tile (the last shard takes the rest)
static void shard_rows(const split_buft_ctx * c, int64_t nrows, int i, int64_t * lo, int64_t * hi) {
    auto edge = [&](int k) -> int64_t { if (k <= 0) return 0; if (k >= c->n_shards) return nrows; int64_t r = (int64_t)((double)nrows * c->bound[k]); r -= r % 128; return r < 0 ? 0 : (r > nrows ? nrows : r); };
    *lo = edge(i); *hi = edge(i + 1);
    if (*hi < *lo) *hi = *lo;
}

// K range [lo, hi) of shard i: fractions of K rounded DOWN to whole 256-weight superblocks (every format's GEMM granule; the last shard takes the rest)
static void shard_k(const split_buft_ctx * c, int64_t K, int i, int64_t * lo, int64_t * hi) {
    auto edge = [&](int k) -> int64_t { if (k <= 0) return 0; if (k >= c->n_shards) return K; int64_t r = (int64_t)((double)K * c->bound[k]); r -= r % 256; return r < 0 ? 0 : (r > K ? K : r); };
    *lo = edge(i); *hi = edge(i + 1);
    if (*hi < *lo) *hi = *lo;
}

static const char * split_buft_get_name(ggml_backend_buffer_type_t buft) { return ((split_buft_ctx *)buft->context)->name.c_str(); }
bool cdna4_buft_is_split(ggml_backend_buffer_type_t buft) { return buft && buft->iface.get_name == split_buft_get_name; }

// ---- buffer
static void split_buffer_free(ggml_backend_buffer_t buffer) {
    split_buffer_ctx * ctx = (split_buffer_ctx *)buffer->context;
    for (split_tensor * t : ctx->tensors) {
        for (int i = 0; i < t->n; i++) if (t->data[i]) { HIP_OK(hipSetDevice(t->dev[i])); HIP_OK(hipDeviceSynchronize()); HIP_OK(hipFree(t->data[i])); }
        delete t;
    }
    delete ctx;
}
static void * split_buffer_get_base(ggml_backend_buffer_t) { return (void *)0x1000; }   // never dereferenced: the allocators only do arithmetic on it
static void split_buffer_init_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor) {
    GGML_ASSERT(tensor->view_src == nullptr);                       // views of split tensors are not supported (as in the reference)
    GGML_ASSERT(ggml_is_contiguous(tensor) && tensor->ne[2] == 1 && tensor->ne[3] == 1);
    split_buffer_ctx * ctx = (split_buffer_ctx *)buffer->context;
    const split_buft_ctx * bc = (const split_buft_ctx *)buffer->buft->context;
    split_tensor * st = new split_tensor;
    st->n = bc->n_shards; st->ksplit = bc->ksplit;
    const size_t row_bytes = ggml_row_size(tensor->type, tensor->ne[0]);
    // K split: shard edges are multiples of 256 (shard_k), the last shard runs to K — for the 32-weight block formats a K that is not a multiple of 256 leaves that shard a
    // ragged (multiple-of-32) tail, which ggml_cdna4_mul_mat takes; nothing to refuse here (this was a process-aborting GGML_ASSERT)
    for (int i = 0; i < st->n; i++) {
        if (bc->ksplit) shard_k(bc, tensor->ne[0], i, &st->lo[i], &st->hi[i]); else shard_rows(bc, tensor->ne[1], i, &st->lo[i], &st->hi[i]);
        st->dev[i] = bc->shard_dev[i]; st->data[i] = nullptr;
        const int64_t span = st->hi[i] - st->lo[i];
        if (span == 0) continue;
        HIP_OK(hipSetDevice(st->dev[i]));
        const size_t bytes = bc->ksplit ? (size_t)tensor->ne[1] * ggml_row_size(tensor->type, span) : (size_t)span * row_bytes;
        HIP_OK(hipMalloc(&st->data[i], bytes + 256));              // slack: kernels read whole 16-byte pieces
    }
    ctx->tensors.push_back(st);
    tensor->extra = st;
}
static void split_buffer_set_tensor(ggml_backend_buffer_t, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor));         // split tensors are written whole
    const split_tensor * st = (const split_tensor *)tensor->extra;
    const size_t row_bytes = tensor->nb[1];
    for (int i = 0; i < st->n; i++) {
        const int64_t rows = st->hi[i] - st->lo[i];
        if (rows == 0) continue;
        HIP_OK(hipSetDevice(st->dev[i]));
        if (st->ksplit) {   // columns [lo, hi) of every row: a byte range of each block-quantized row (blocks run along K inside a row)
            const size_t off = ggml_row_size(tensor->type, st->lo[i]), wid = ggml_row_size(tensor->type, rows);
            HIP_OK(hipMemcpy2D(st->data[i], wid, (const char *)data + off, row_bytes, wid, (size_t)tensor->ne[1], hipMemcpyHostToDevice));
        } else HIP_OK(hipMemcpy(st->data[i], (const char *)data + st->lo[i] * row_bytes, (size_t)rows * row_bytes, hipMemcpyHostToDevice));
    }
}
static void split_buffer_get_tensor(ggml_backend_buffer_t, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor));
    const split_tensor * st = (const split_tensor *)tensor->extra;
    const size_t row_bytes = tensor->nb[1];
    for (int i = 0; i < st->n; i++) {
        const int64_t rows = st->hi[i] - st->lo[i];
        if (rows == 0) continue;
        HIP_OK(hipSetDevice(st->dev[i]));
        if (st->ksplit) {
            const size_t off = ggml_row_size(tensor->type, st->lo[i]), wid = ggml_row_size(tensor->type, rows);
            HIP_OK(hipMemcpy2D((char *)data + off, row_bytes, st->data[i], wid, wid, (size_t)tensor->ne[1], hipMemcpyDeviceToHost));
        } else HIP_OK(hipMemcpy((char *)data + st->lo[i] * row_bytes, st->data[i], (size_t)rows * row_bytes, hipMemcpyDeviceToHost));
    }
}
static void split_buffer_clear(ggml_backend_buffer_t, uint8_t) {}
static const ggml_backend_buffer_i split_buffer_iface = {
    /* .free_buffer   = */ split_buffer_free,
    /* .get_base      = */ split_buffer_get_base,
    /* .init_tensor   = */ split_buffer_init_tensor,
    /* .memset_tensor = */ NULL,
    /* .set_tensor    = */ split_buffer_set_tensor,
    /* .get_tensor    = */ split_buffer_get_tensor,
    /* .cpy_tensor    = */ NULL,
    /* .clear         = */ split_buffer_clear,
    /* .reset         = */ NULL,
};

// ---- buffer type
static ggml_backend_buffer_t split_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    // the per-device slices are allocated tensor by tensor in init_tensor (the split is only known per tensor)
    return ggml_backend_buffer_init(buft, split_buffer_iface, new split_buffer_ctx, size);
}
static size_t split_buft_get_alignment(ggml_backend_buffer_type_t) { return 256; }
static size_t split_buft_get_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * tensor) { return (ggml_nbytes(tensor) + 255) & ~(size_t)255; }
static bool split_buft_is_host(ggml_backend_buffer_type_t) { return false; }
static const ggml_backend_buffer_type_i split_buft_iface = {
    /* .get_name       = */ split_buft_get_name,
    /* .alloc_buffer   = */ split_buft_alloc_buffer,
    /* .get_alignment  = */ split_buft_get_alignment,
    /* .get_max_size   = */ NULL,
    /* .get_alloc_size = */ split_buft_get_alloc_size,
    /* .is_host        = */ split_buft_is_host,
};

namespace { bool rccl_ready(int ndev); }                            // (below)
// cumulative fractions and shard devices of a split over c->n_shards shards (NULL / all-zero tensor_split: equal shares; self-sharding: every shard on the main device)
static void fill_bounds(split_buft_ctx * c, const float * tensor_split) {
    float w[CDNA4_MAX_DEVICES]; float sum = 0.f;
    for (int i = 0; i < c->n_shards; i++) { w[i] = (tensor_split && !c->self) ? tensor_split[i] : 0.f; if (w[i] < 0.f) w[i] = 0.f; sum += w[i]; }
    if (sum <= 0.f) { for (int i = 0; i < c->n_shards; i++) w[i] = 1.f; sum = (float)c->n_shards; }      // NULL / all zero: equal shares
    c->bound[0] = 0.f;
    for (int i = 0; i < c->n_shards; i++) { c->bound[i + 1] = c->bound[i] + w[i] / sum; c->shard_dev[i] = c->self ? c->main_device : i; }
    c->bound[c->n_shards] = 1.f;
}
// The partition a split buffer type of `n_devices` devices gives a tensor with `n` rows (row split) or K = `n` (K split): lo / hi / dev per shard.  Pure host arithmetic —
// the SAME functions init_tensor uses — exported (symbol + get_proc_address) so that hosts can plan a layout, and so that the partition is testable without a multi-GPU
// node (tests/test_split_ranges.py: main_device != 0, uneven and zero shares, ragged K).  Returns the number of shards, or -1 for bad arguments.
extern "C" __attribute__((visibility("default"))) int ggml_backend_cdna4_split_ranges(int ksplit, int main_device, int n_devices, const float * tensor_split, int64_t n,
                                                                                     int64_t * lo, int64_t * hi, int * dev) {
    if (n_devices < 1 || n_devices > CDNA4_MAX_DEVICES || main_device < 0 || main_device >= n_devices || n < 0 || !lo || !hi) return -1;
    split_buft_ctx c{};
    c.main_device = main_device; c.ksplit = ksplit != 0; c.self = false; c.n_shards = n_devices;
    fill_bounds(&c, tensor_split);
    for (int i = 0; i < c.n_shards; i++) {
        if (c.ksplit) shard_k(&c, n, i, &lo[i], &hi[i]); else shard_rows(&c, n, i, &lo[i], &hi[i]);
        if (dev) dev[i] = c.shard_dev[i];
    }
    return c.n_shards;
}
static ggml_backend_buffer_type_t make_split_buffer_type(int main_device, const float * tensor_split, bool ksplit) {
    static std::mutex mu;
    static std::map<std::string, ggml_backend_buffer_type *> cache;     // one buffer type per distinct (main device, split)
    std::lock_guard<std::mutex> lock(mu);
    const int ndev = cdna4_reg_device_count();
    if (main_device < 0 || main_device >= ndev) return nullptr;
    split_buft_ctx c{};
    c.main_device = main_device; c.ksplit = ksplit;
    const int self = self_shards();
    c.self = self > 1;
    c.n_shards = c.self ? self : ndev;
    fill_bounds(&c, tensor_split);
    std::string key = std::to_string(main_device) + (c.self ? "s" : "d") + (ksplit ? "k" : "r");
    for (int i = 0; i <= c.n_shards; i++) key += ":" + std::to_string(c.bound[i]);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    c.name = ksplit ? "CDNA4_KSplit" : "CDNA4_Split";
    // the K split's collective: communicators are created HERE, when the host asks for the buffer type (model load), not inside the first graph_compute (a multi-second
    // stall there: ADVICE r4).  EXPERIMENTAL until it has run on a node: the multi-rank exchange has only ever executed as a world of one (no multi-GPU box in any round's pool);
    // GGML_CDNA4_KSPLIT_RCCL=0 keeps the peer-copy + ggml_cdna4_sum_partials path, which is what one-GPU boxes and GGML_CDNA4_SPLIT_SELF exercise.
    if (ksplit && !c.self && ndev > 1 && !(getenv("GGML_CDNA4_KSPLIT_RCCL") && atoi(getenv("GGML_CDNA4_KSPLIT_RCCL")) == 0)) (void)rccl_ready(ndev);
    ggml_backend_buffer_type * buft = new ggml_backend_buffer_type{ /* .iface = */ split_buft_iface, /* .device = */ cdna4_reg_device(main_device), /* .context = */ new split_buft_ctx(c) };
    cache[key] = buft;
    return buft;
}

ggml_backend_buffer_type_t cdna4_split_buffer_type(int main_device, const float * tensor_split) { return make_split_buffer_type(main_device, tensor_split, false); }
ggml_backend_buffer_type_t cdna4_ksplit_buffer_type(int main_device, const float * tensor_split) { return make_split_buffer_type(main_device, tensor_split, true); }

// ---- RCCL, loaded at run time (the plug-in has no link-time dependency on it; absent or failing -> peer copies + ggml_cdna4_sum_partials)
namespace {
typedef void * nccl_comm_t;
struct rccl_api {
    bool tried = false, ok = false;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, nccl_comm_t, hipStream_t) = nullptr;       // rccl.h ncclReduce(send, recv, count, type, op, root, comm, stream)
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
    nccl_comm_t comms[CDNA4_MAX_DEVICES] = {}; int ncomm = 0;
};
rccl_api g_rccl;
std::mutex g_rccl_mu;
// communicators over devices 0 .. ndev-1 (rccl.h:236 ncclCommInitAll: one process driving all GPUs of the node)
bool rccl_ready(int ndev) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.tried) return g_rccl.ok && g_rccl.ncomm == ndev;
    g_rccl.tried = true;
    if (getenv("GGML_CDNA4_NO_RCCL") && atoi(getenv("GGML_CDNA4_NO_RCCL")) != 0) return false;
    void * h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    g_rccl.CommInitAll = (int (*)(nccl_comm_t *, int, const int *))dlsym(h, "ncclCommInitAll");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart"); g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    g_rccl.GetErrorString = (const char * (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.Reduce = (int (*)(const void *, void *, size_t, int, int, int, nccl_comm_t, hipStream_t))dlsym(h, "ncclReduce");
    g_rccl.CommDestroy = (int (*)(nccl_comm_t))dlsym(h, "ncclCommDestroy");
    if (!g_rccl.CommInitAll || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) return false;
    int devs[CDNA4_MAX_DEVICES];
    for (int i = 0; i < ndev; i++) devs[i] = i;
    int cur = 0; (void)hipGetDevice(&cur);
    const int rc = g_rccl.CommInitAll(g_rccl.comms, ndev, devs);
    (void)hipSetDevice(cur);
    if (rc != 0) { fprintf(stderr, "ggml-cdna4: ncclCommInitAll failed (%s): K-split falls back to peer copies\n", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"); return false; }
    g_rccl.ncomm = ndev; g_rccl.ok = true;
    atexit([] {                                                          // the communicators live as long as the process (buffer types are cached for its lifetime): released at exit
        if (!g_rccl.CommDestroy) return;
        for (int i = 0; i < g_rccl.ncomm; i++) if (g_rccl.comms[i]) { (void)g_rccl.CommDestroy(g_rccl.comms[i]); g_rccl.comms[i] = nullptr; }
    });
    return true;
}
}

// ---- MUL_MAT with row-split weights
bool cdna4_split_supports_mul_mat(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    if (a->ne[2] != 1 || a->ne[3] != 1 || !a->extra) return false;                  // one matrix, initialised by init_tensor
    if (((const split_tensor *)a->extra)->ksplit && !ggml_is_quantized(a->type)) return false;                         // K split: the block-quantized formats (whole superblocks per shard)
    if (b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !ggml_is_contiguous(b) || !ggml_is_contiguous(op)) return false;
    return ggml_cdna4_row_size((int)a->type, a->ne[0]) != 0 && ggml_cdna4_mul_mat_workspace_size((int)a->type, a->ne[0], 1) != 0;
}
static void * lane_buf(void ** p, size_t * have, size_t want) {
    if (want <= *have) return *p;
    if (*p) HIP_OK(hipFree(*p));
    *have = (want + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    if (hipMalloc(p, *have) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; *have = 0; }
    return *p;
}
void cdna4_split_free_lanes(cdna4_backend_ctx * ctx) {
    for (cdna4_lane & l : ctx->lanes) {
        if (l.device < 0) continue;
        (void)hipSetDevice(l.device);
        if (l.stream) { (void)hipStreamSynchronize(l.stream); if (!l.shares_stream) (void)hipStreamDestroy(l.stream); }
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.x) (void)hipFree(l.x); if (l.y) (void)hipFree(l.y); if (l.ws) (void)hipFree(l.ws);
        if (l.ym) { (void)hipSetDevice(ctx->device); (void)hipFree(l.ym); }
        l = cdna4_lane{};
    }
    if (ctx->ev_x) { (void)hipSetDevice(ctx->device); (void)hipEventDestroy(ctx->ev_x); ctx->ev_x = nullptr; }
}

// the lane (stream, event, staging buffers) of shard i on device dev
static cdna4_lane & lane_of(cdna4_backend_ctx * ctx, int i, int dev) {
    cdna4_lane & l = ctx->lanes[i];
    if (l.device < 0) {
        l.device = dev;
        // ONE stream per device: the kernel library keeps its split-K exchange scratch per device, so GEMM calls for one device
        // must be stream-ordered (include/ggml_cdna4.h) — shards that share a device (GGML_CDNA4_SPLIT_SELF) share a stream
        for (int j = 0; j < CDNA4_MAX_DEVICES && !l.stream; j++) if (j != i && ctx->lanes[j].device == dev && ctx->lanes[j].stream) { l.stream = ctx->lanes[j].stream; l.shares_stream = true; }
        if (!l.stream) HIP_OK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
        if (dev != ctx->device) { const hipError_t e = hipDeviceEnablePeerAccess(ctx->device, 0); if (e != hipSuccess) (void)hipGetLastError(); }   // xGMI peer mapping (already enabled: fine)
    }
    return l;
}

// ---- MUL_MAT with K-split weights: per shard one ggml_cdna4_mul_mat over its K range -> a full-size partial [B][M]; then the sum
static enum ggml_status ksplit_mul_mat(cdna4_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * a = dst->src[0], * b = dst->src[1];
    const split_tensor * st = (const split_tensor *)a->extra;
    const split_buft_ctx * bc = (const split_buft_ctx *)a->buffer->buft->context;
    const int64_t K = a->ne[0], M = a->ne[1], B = b->ne[1] * b->ne[2] * b->ne[3];
    const size_t y_bytes = (size_t)B * M * sizeof(float);
    HIP_OK(hipSetDevice(ctx->device));
    if (!ctx->ev_x) HIP_OK(hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming));
    HIP_OK(hipEventRecord(ctx->ev_x, ctx->stream));
    int active[CDNA4_MAX_DEVICES], n_active = 0; bool every_device = !bc->self;
    for (int i = 0; i < st->n; i++) { if (st->hi[i] > st->lo[i]) active[n_active++] = i; else every_device = false; }
    if (n_active == 0) { HIP_OK(hipMemsetAsync(dst->data, 0, y_bytes, ctx->stream)); return GGML_STATUS_SUCCESS; }
    // one RCCL all-reduce when every device of the node holds a shard (the communicator spans them all); GGML_CDNA4_KSPLIT_RCCL=0 forces the fall-back
    static const bool rccl_off = getenv("GGML_CDNA4_KSPLIT_RCCL") && atoi(getenv("GGML_CDNA4_KSPLIT_RCCL")) == 0;
    const bool use_rccl = every_device && !rccl_off && rccl_ready(st->n);
    enum ggml_status status = GGML_STATUS_SUCCESS;
    for (int q = 0; q < n_active && status == GGML_STATUS_SUCCESS; q++) {
        const int i = active[q], dev = st->dev[i];
        const int64_t klo = st->lo[i], klen = st->hi[i] - st->lo[i];
        const size_t ws_need = ggml_cdna4_mul_mat_workspace_size((int)a->type, klen, B);
        HIP_OK(hipSetDevice(dev));
        cdna4_lane & l = lane_of(ctx, i, dev);
        if (!lane_buf(&l.y, &l.y_bytes, y_bytes) || !lane_buf(&l.ws, &l.ws_bytes, ws_need)) { status = GGML_STATUS_ALLOC_FAILED; break; }
        HIP_OK(hipStreamWaitEvent(l.stream, ctx->ev_x, 0));
        const float * xp = (const float *)b->data + klo; int64_t xs = K;        // same device: the K range of X in place (the kernel library takes a row stride)
        if (dev != ctx->device) {                                                 // another GPU: its K columns of X over xGMI, densely packed
            if (!lane_buf(&l.x, &l.x_bytes, (size_t)B * klen * sizeof(float))) { status = GGML_STATUS_ALLOC_FAILED; break; }
            HIP_OK(hipMemcpy2DAsync(l.x, (size_t)klen * sizeof(float), xp, (size_t)K * sizeof(float), (size_t)klen * sizeof(float), (size_t)B, hipMemcpyDeviceToDevice, l.stream));
            xp = (const float *)l.x; xs = klen;
        }
        if (ggml_cdna4_mul_mat((int)a->type, st->data[i], (int64_t)ggml_row_size(a->type, klen), xp, xs, (float *)l.y, M, M, klen, B,
                               l.ws, l.ws_bytes, GGML_CDNA4_PATH_AUTO, 0, 0, l.stream)) { fprintf(stderr, "ggml-cdna4: K-split MUL_MAT failed: %s\n", ggml_cdna4_last_error()); status = GGML_STATUS_FAILED; break; }
    }
    if (status != GGML_STATUS_SUCCESS) { HIP_OK(hipSetDevice(ctx->device)); return status; }
    if (use_rccl) {
        // only the main device consumes the sum: ncclReduce to its rank (rank = device index: ncclCommInitAll over devices 0 .. n-1), one call per device inside a group —
        // 1/n of an all-reduce's incoming traffic on every other device; ncclAllReduce where the library lacks the symbol
        int rc = g_rccl.GroupStart();
        for (int q = 0; q < n_active && rc == 0; q++) {
            const int i = active[q], dev = st->dev[i];
            HIP_OK(hipSetDevice(dev));
            cdna4_lane & l = ctx->lanes[i];
            if (g_rccl.Reduce) rc = g_rccl.Reduce(l.y, dev == ctx->device ? dst->data : l.y, (size_t)B * M, /* ncclFloat32 */ 7, /* ncclSum */ 0, /* root */ ctx->device, g_rccl.comms[dev], l.stream);
            else rc = g_rccl.AllReduce(l.y, dev == ctx->device ? dst->data : l.y, (size_t)B * M, /* ncclFloat32 */ 7, /* ncclSum */ 0, g_rccl.comms[dev], l.stream);
        }
        const int rc2 = g_rccl.GroupEnd();
        if (rc != 0 || rc2 != 0) { fprintf(stderr, "ggml-cdna4: ncclAllReduce failed (%s)\n", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc ? rc : rc2) : "?"); HIP_OK(hipSetDevice(ctx->device)); return GGML_STATUS_FAILED; }
        for (int q = 0; q < n_active; q++) { const int i = active[q]; HIP_OK(hipSetDevice(st->dev[i])); HIP_OK(hipEventRecord(ctx->lanes[i].done, ctx->lanes[i].stream)); }
        HIP_OK(hipSetDevice(ctx->device));
        for (int q = 0; q < n_active; q++) HIP_OK(hipStreamWaitEvent(ctx->stream, ctx->lanes[active[q]].done, 0));
        ctx->n_ksplit_rccl++;
        return GGML_STATUS_SUCCESS;
    }
    // fall-back: the partials meet on the main device and are added there in shard order (deterministic)
    const float * parts[CDNA4_MAX_DEVICES];
    for (int q = 0; q < n_active; q++) {
        const int i = active[q], dev = st->dev[i];
        cdna4_lane & l = ctx->lanes[i];
        parts[q] = (const float *)l.y;
        if (dev != ctx->device) {
            HIP_OK(hipSetDevice(ctx->device));
            if (!lane_buf(&l.ym, &l.ym_bytes, y_bytes)) return GGML_STATUS_ALLOC_FAILED;
            HIP_OK(hipSetDevice(dev));
            HIP_OK(hipMemcpyPeerAsync(l.ym, ctx->device, l.y, dev, y_bytes, l.stream));
            parts[q] = (const float *)l.ym;
        }
        HIP_OK(hipSetDevice(dev));
        HIP_OK(hipEventRecord(l.done, l.stream));
    }
    HIP_OK(hipSetDevice(ctx->device));
    for (int q = 0; q < n_active; q++) HIP_OK(hipStreamWaitEvent(ctx->stream, ctx->lanes[active[q]].done, 0));
    ctx->n_ksplit_sum++;
    if (ggml_cdna4_sum_partials((float *)dst->data, parts, n_active, B * M, ctx->stream)) { fprintf(stderr, "ggml-cdna4: K-split sum failed: %s\n", ggml_cdna4_last_error()); return GGML_STATUS_FAILED; }
    return GGML_STATUS_SUCCESS;
}

enum ggml_status cdna4_split_mul_mat(cdna4_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * a = dst->src[0], * b = dst->src[1];
    const split_tensor * st = (const split_tensor *)a->extra;
    if (st->ksplit) return ksplit_mul_mat(ctx, dst);
    const split_buft_ctx * bc = (const split_buft_ctx *)a->buffer->buft->context;
    const int64_t K = a->ne[0], M = a->ne[1], B = b->ne[1] * b->ne[2] * b->ne[3];       // contiguous b: all activation rows form one matrix
    const size_t x_bytes = (size_t)B * K * sizeof(float);
    const size_t ws_need = ggml_cdna4_mul_mat_workspace_size((int)a->type, K, B);
    HIP_OK(hipSetDevice(ctx->device));
    if (!ctx->ev_x) HIP_OK(hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming));
    HIP_OK(hipEventRecord(ctx->ev_x, ctx->stream));                                     // everything that produced X precedes this point
    enum ggml_status status = GGML_STATUS_SUCCESS;
    for (int i = 0; i < st->n && status == GGML_STATUS_SUCCESS; i++) {
        const int64_t rows = st->hi[i] - st->lo[i];
        if (rows == 0) continue;
        const int dev = st->dev[i];
        const bool local = dev == ctx->device && !bc->self;
        if (local) {                                                                    // the main device's shard: straight into dst, on the main stream
            HIP_OK(hipSetDevice(ctx->device));                                          // (a remote shard in front of it left ITS device current: main_device > 0)
            void * ws = ctx->need_ws(ws_need);
            if (!ws) return GGML_STATUS_ALLOC_FAILED;
            if (ggml_cdna4_mul_mat((int)a->type, st->data[i], (int64_t)a->nb[1], (const float *)b->data, K, (float *)dst->data + st->lo[i], M, rows, K, B,
                                   ws, ctx->ws_size, GGML_CDNA4_PATH_AUTO, 0, 0, ctx->stream)) { fprintf(stderr, "ggml-cdna4: split MUL_MAT failed: %s\n", ggml_cdna4_last_error()); status = GGML_STATUS_FAILED; }
            continue;
        }
        HIP_OK(hipSetDevice(dev));
        cdna4_lane & l = lane_of(ctx, i, dev);
        if (!lane_buf(&l.x, &l.x_bytes, x_bytes) || !lane_buf(&l.y, &l.y_bytes, (size_t)B * rows * sizeof(float)) || !lane_buf(&l.ws, &l.ws_bytes, ws_need)) { status = GGML_STATUS_ALLOC_FAILED; break; }
        HIP_OK(hipStreamWaitEvent(l.stream, ctx->ev_x, 0));
        if (dev != ctx->device) HIP_OK(hipMemcpyPeerAsync(l.x, dev, b->data, ctx->device, x_bytes, l.stream));
        else HIP_OK(hipMemcpyAsync(l.x, b->data, x_bytes, hipMemcpyDeviceToDevice, l.stream));
        if (ggml_cdna4_mul_mat((int)a->type, st->data[i], (int64_t)a->nb[1], (const float *)l.x, K, (float *)l.y, rows, rows, K, B,
                               l.ws, l.ws_bytes, GGML_CDNA4_PATH_AUTO, 0, 0, l.stream)) { fprintf(stderr, "ggml-cdna4: split MUL_MAT failed: %s\n", ggml_cdna4_last_error()); status = GGML_STATUS_FAILED; break; }
        // the shard's [B][rows] block into columns [lo, hi) of dst's [B][M] rows (a strided copy over xGMI when dev != main)
        HIP_OK(hipMemcpy2DAsync((float *)dst->data + st->lo[i], (size_t)M * sizeof(float), l.y, (size_t)rows * sizeof(float), (size_t)rows * sizeof(float), (size_t)B,
                                hipMemcpyDeviceToDevice, l.stream));
        HIP_OK(hipEventRecord(l.done, l.stream));
    }
    HIP_OK(hipSetDevice(ctx->device));
    for (int i = 0; i < st->n; i++) if (ctx->lanes[i].device >= 0 && st->hi[i] > st->lo[i] && !(st->dev[i] == ctx->device && !bc->self)) HIP_OK(hipStreamWaitEvent(ctx->stream, ctx->lanes[i].done, 0));
    return status;
}
