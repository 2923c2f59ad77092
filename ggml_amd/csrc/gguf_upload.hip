// gguf_upload.hip — GGUF tensor payloads from the read-only mapping (gguf_reader.cpp) into HBM.
//
// NOT yet run on a GPU (written in a session without one; its test is opt-in, CDNA4_TEST_EXPERIMENTAL=1).
// The mapping is pageable file memory, which the DMA engines cannot read directly, so the bytes go through two pinned
// staging buffers: while buffer A's chunk is in flight to the device (hipMemcpyAsync on the caller's stream), the CPU
// copies the next chunk of the file into buffer B; an event per buffer says when it may be refilled.  Reference
// counterpart: the read of the data blob in gguf_init_from_file (src/gguf.cpp:644-660) followed by
// ggml_backend_tensor_set per tensor (examples/gpt-2/main-backend.cpp:412-420) — two host copies and a blocking upload there.
#include "cdna4_common.h"
#include "../../include/ggml_cdna4_gguf.h"
#include <string.h>

namespace {
constexpr size_t kChunk = 16u << 20;             // 16 MiB per staging buffer
struct Stage {
    void *buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false};
    int ensure() {
        for (int i = 0; i < 2; i++) {
            if (!buf[i]) { hipError_t e = hipHostMalloc(&buf[i], kChunk, hipHostMallocDefault); if (e != hipSuccess) { buf[i] = nullptr; return cdna4_set_error(e, __FILE__, __LINE__); } }
            if (!ev[i]) { hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); if (e != hipSuccess) { ev[i] = nullptr; return cdna4_set_error(e, __FILE__, __LINE__); } }
        }
        return 0;
    }
};
thread_local Stage g_stage;                      // one pair per host thread (a ggml_backend_t is driven by one thread at a time)

int upload_range(const uint8_t *src, size_t n, uint8_t *dst, hipStream_t st) {
    if (g_stage.ensure() != 0) return -1;
    int cur = 0;
    for (size_t off = 0; off < n; off += kChunk, cur ^= 1) {
        const size_t len = n - off < kChunk ? n - off : kChunk;
        if (g_stage.busy[cur]) { hipError_t e = hipEventSynchronize(g_stage.ev[cur]); if (e != hipSuccess) return cdna4_set_error(e, __FILE__, __LINE__); g_stage.busy[cur] = false; }
        memcpy(g_stage.buf[cur], src + off, len);                                  // page-cache / disk read happens here
        hipError_t e = hipMemcpyAsync(dst + off, g_stage.buf[cur], len, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return cdna4_set_error(e, __FILE__, __LINE__);
        e = hipEventRecord(g_stage.ev[cur], st);
        if (e != hipSuccess) return cdna4_set_error(e, __FILE__, __LINE__);
        g_stage.busy[cur] = true;
    }
    // the staging buffers may be reused by the next call: nothing may still be reading them (the device copy itself stays
    // asynchronous to later work on `st` only up to this point)
    for (int i = 0; i < 2; i++) if (g_stage.busy[i]) { hipError_t e = hipEventSynchronize(g_stage.ev[i]); if (e != hipSuccess) return cdna4_set_error(e, __FILE__, __LINE__); g_stage.busy[i] = false; }
    return 0;
}
}  // namespace

extern "C" {

int ggml_cdna4_gguf_upload(const ggml_cdna4_gguf *g, int64_t tensor_id, void *dst_device, size_t dst_bytes, void *stream) {
    const void *src = ggml_cdna4_gguf_tensor_data(g, tensor_id);
    if (!src) return -1;                                                           // message set by tensor_data
    const size_t n = ggml_cdna4_gguf_tensor_size(g, tensor_id);
    if (!dst_device || dst_bytes < n) return cdna4_set_error_msg("gguf_upload: destination buffer too small");
    return upload_range((const uint8_t *)src, n, (uint8_t *)dst_device, (hipStream_t)stream);
}

}  // extern "C"
