// ggml_cdna4_ops.h — plug-in-internal: maps the non-MUL_MAT ggml ops (and F32/F16 MUL_MAT) onto the plain-pointer
// kernels of libcdna4_kernels.so (include/ggml_cdna4.h, "supporting ops" section).
#pragma once
#include "ggml.h"
#include "ggml-backend.h"

// can the device run this node?  (called from supports_op for everything except quantized MUL_MAT / MUL_MAT_ID)
bool cdna4_ops_supports_tensor(const ggml_tensor * op);
// F32 / F16 weights MUL_MAT
bool cdna4_ops_supports_matmul(const ggml_tensor * op);
// run one node on the backend's stream; backend_ctx is the opaque cdna4_backend_ctx
enum ggml_status cdna4_ops_compute(void * backend_ctx, ggml_tensor * node);

// GGML_CDNA4_EXACT=1 (reference-order kernels of exact.hip; peepholes and HIP-graph replay stay off)
bool cdna4_exact_mode();

extern "C" void * cdna4_backend_stream(void * backend_ctx);
extern "C" void * cdna4_backend_scratch(void * backend_ctx, size_t nbytes);
