// ggml_cdna4_internal.h — structures shared by the translation units of the plug-in (ggml_cdna4_backend.cpp: the five vtables;
// ggml_cdna4_split.cpp: the row-split buffer type and its multi-device MUL_MAT).
#pragma once
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml_cdna4.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define CDNA4_MAX_DEVICES 16
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "ggml-cdna4: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); GGML_ABORT("HIP error"); } } while (0)

// one shard of a row-split MUL_MAT on another device (or, under GGML_CDNA4_SPLIT_SELF, another slice of the same one): its own
// stream, scratch and staging buffers, created on first use and kept for the life of the backend
struct cdna4_lane {
    int device = -1;
    hipStream_t stream = nullptr; bool shares_stream = false;   // (shards on one device share its stream)
    hipEvent_t done = nullptr;
    void * x = nullptr;  size_t x_bytes = 0;      // activations copied from the main device
    void * y = nullptr;  size_t y_bytes = 0;      // this shard's output rows, [B][rows] f32
    void * ws = nullptr; size_t ws_bytes = 0;     // ggml_cdna4_mul_mat workspace
    void * ym = nullptr; size_t ym_bytes = 0;     // K split without RCCL: this shard's partial [B][M], copied to the MAIN device for the sum
};

#define CDNA4_GRAPH_SLOTS 4
struct cdna4_backend_ctx {
    int device; hipStream_t stream; std::string name;
    void * ws = nullptr; size_t ws_size = 0;
    hipEvent_t ev_x = nullptr;                    // "activations are ready on the main stream" (split MUL_MAT)
    hipEvent_t ev_copy = nullptr;                 // cpy_tensor_async between two backends
    cdna4_lane lanes[CDNA4_MAX_DEVICES];
    // HIP-graph replay of a compute graph that comes back unchanged (ggml_cdna4_backend.cpp: graph_compute)
    // a few graphs at a time (a ggml_backend_sched graph reaches one backend as several splits, in turn), least recently used replaced
    struct graph_slot {
        uint64_t sig = 0;                                     // graph_signature of the graph seen (0: slot unused)
        uint64_t gen = 0;                                     // library-scratch generation + this context's workspace generation at capture time
        uint64_t last_use = 0;
        hipGraphExec_t exec = nullptr;                        // null: seen once, not captured yet
    } graph_slots[CDNA4_GRAPH_SLOTS];
    uint64_t graph_clock = 0;
    uint64_t ws_gen = 0;
    bool graphs_off = false;                                  // a capture failed once: stay on plain launches
    int n_graph_launches = 0, n_graph_captures = 0;           // (statistics, printed at free under GGML_CDNA4_STATS)
    int n_ksplit_rccl = 0, n_ksplit_sum = 0;                  // K-split MUL_MATs reduced by an RCCL all-reduce / by the in-order sum
    // the quantized activations the workspace holds (ggml_cdna4_act_image_key; ggml_cdna4_backend.cpp: act_image_*): the next MUL_MAT of the same src1 multiplies them
    // without quantizing again.  `uses` = ws_uses when they were written: any later hand-out of the workspace (another op's scratch, a growth) ends their life
    uint64_t ws_uses = 0;
    struct { const void * x = nullptr; int64_t x_stride = 0, K = 0, B = 0; size_t x_bytes = 0; uint32_t key = 0; uint64_t uses = 0;
             const void * producer = nullptr; } act_image;     // producer: the node whose own launch wrote x AND the image (NORM chain: its write of x does not end the image's life)
    int n_act_produced = 0;                                   // NORM chains that left the image of their rows for the next MUL_MAT
    int n_act_shared = 0;                                     // MUL_MATs that took the hand-off (statistics; "ggml_backend_cdna4_act_shared_count")
    // the FRONT a prefill-sized MUL_MAT_ID left in the workspace (ggml_cdna4_mul_mat_id_front_key: sorted ids, tile records, spans, quantized activations): the next MUL_MAT_ID of
    // the same (b, ids) and shape — the gate stack behind the up stack of a mixture-of-experts layer — multiplies it again.  Same life as act_image.
    struct { uint32_t key = 0; uint64_t uses = 0; const void * b = nullptr, * ids = nullptr; size_t b_bytes = 0, ids_bytes = 0; int type = 0;
             int64_t b_row = 0, b_tok = 0, ids_tok = 0, M = 0, K = 0, n_expert = 0, n_used = 0, n_b = 0, n_tok = 0; } moe_front;
    int n_moe_front_shared = 0;                               // MUL_MAT_IDs that multiplied the previous one's front (statistics; "ggml_backend_cdna4_moe_front_shared_count")
    int n_grouped = 0;                                        // one-row MUL_MATs that rode in another one's launch (statistics; "ggml_backend_cdna4_grouped_count")
    void * need_ws(size_t n) {
        ws_uses++;
        if (n <= ws_size) return ws;
        HIP_OK(hipStreamSynchronize(stream));
        ws_gen++;                                             // a captured graph holds the old workspace address
        if (ws) HIP_OK(hipFree(ws));
        ws_size = (n + (8u << 20)) & ~(size_t)((1u << 20) - 1);
        if (hipMalloc(&ws, ws_size) != hipSuccess) { (void)hipGetLastError(); ws = nullptr; ws_size = 0; }
        return ws;
    }
};

// ---- ggml_cdna4_split.cpp
// the well-known proc address "ggml_backend_split_buffer_type" (include/ggml-backend.h:188): weights whose rows (output features)
// are sharded over the node's GPUs; tensor_split = CDNA4_MAX_DEVICES proportions (NULL / all zero: equal shares)
ggml_backend_buffer_type_t cdna4_split_buffer_type(int main_device, const float * tensor_split);
// "ggml_backend_cdna4_ksplit_buffer_type" (same signature; no reference counterpart): the reduction dimension K sharded instead of the rows, the partial
// outputs summed by one RCCL all-reduce (ggml_cdna4_split.cpp)
ggml_backend_buffer_type_t cdna4_ksplit_buffer_type(int main_device, const float * tensor_split);
// the partition either split buffer type gives a tensor (pure arithmetic; exported as a symbol and through get_proc_address): ggml_cdna4_split.cpp
extern "C" int ggml_backend_cdna4_split_ranges(int ksplit, int main_device, int n_devices, const float * tensor_split, int64_t n, int64_t * lo, int64_t * hi, int * dev);
bool cdna4_buft_is_split(ggml_backend_buffer_type_t buft);
bool cdna4_split_supports_mul_mat(const ggml_tensor * op);
enum ggml_status cdna4_split_mul_mat(cdna4_backend_ctx * ctx, ggml_tensor * dst);
void cdna4_split_free_lanes(cdna4_backend_ctx * ctx);
int cdna4_reg_device_count(void);
ggml_backend_dev_t cdna4_reg_device(int i);
