// TEST INFRASTRUCTURE: a host stand-in for the HIP RUNTIME API as the ggml plug-in (ggml_amd/csrc/backend/*.cpp) uses it, so that the plug-in's host logic — the graph walk, its
// peepholes, the hand-off of quantized activations, the buffer types — can run on a box without a GPU against the whole-library emulation (tools/emul/lib_emul_so.cpp:
// libcdna4_emul.so, the product's kernel sources compiled for the CPU).  "Device" memory is the emulation's shared mappings (the work-group processes of an emulated launch
// write into them); streams and events are tokens (every emulated launch is synchronous); graph capture reports failure, which the plug-in answers by staying on plain launches.
// Nothing under ggml_amd/ includes this file: tools/emul/plugin_emul_check.py puts its directory in front of the include path of a SEPARATE build of the plug-in.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801 };
typedef struct emu_stream_tag * hipStream_t;
typedef struct emu_event_tag * hipEvent_t;
typedef struct emu_graph_tag * hipGraph_t;
typedef struct emu_graphexec_tag * hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterPortable = 1, hipHostRegisterMapped = 2 };
struct hipDeviceProp_t { char name[256]; size_t totalGlobalMem; int multiProcessorCount; char gcnArchName[256]; };

extern "C" void * cdna4_emul_alloc(size_t n);                          // libcdna4_emul.so: a shared mapping between guard pages

static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char * hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated HIP runtime: not supported"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int * d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t * p, int) {
    memset(p, 0, sizeof(*p)); strcpy(p->name, "CDNA4 emulation (CPU)"); strcpy(p->gcnArchName, "gfx950-emulated");
    p->totalGlobalMem = (size_t)8 << 30; const char * e = getenv("EMU_CUS"); p->multiProcessorCount = e ? atoi(e) : 256; return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t * fr, size_t * tot) { *fr = (size_t)6 << 30; *tot = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipMalloc(void ** p, size_t n) { *p = cdna4_emul_alloc(n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void *) { return hipSuccess; }         // (the mappings stay: test processes are short-lived)
static inline hipError_t hipHostMalloc(void ** p, size_t n, unsigned) { *p = cdna4_emul_alloc(n ? n : 1); return hipSuccess; }
static inline hipError_t hipHostFree(void *) { return hipSuccess; }
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void ** d, void * h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipMemset(void * d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void * d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void * d, const void * s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void * d, const void * s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeer(void * d, int, const void * s, int, size_t n) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void * d, int, const void * s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void * d, size_t dp, const void * s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
    for (size_t r = 0; r < h; r++) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void * d, size_t dp, const void * s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
static inline hipError_t hipStreamCreate(hipStream_t * s) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t * s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t * e, unsigned) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
// graph capture is not emulated: the plug-in stays on plain launches (cdna4_backend_graph_compute: "a failed capture turns it off for the backend instance")
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t * g) { *g = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, void *, void *, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
