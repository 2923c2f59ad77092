// ggml_cdna4_ops.cpp — supporting-op dispatch of the plug-in (see ggml_cdna4_ops.h).
#include "ggml_cdna4_ops.h"
#include "ggml_cdna4.h"

bool cdna4_ops_supports_tensor(const ggml_tensor * op) { (void)op; return false; }
bool cdna4_ops_supports_matmul(const ggml_tensor * op) { (void)op; return false; }
enum ggml_status cdna4_ops_compute(void * backend_ctx, ggml_tensor * node) { (void)backend_ctx; (void)node; return GGML_STATUS_FAILED; }
