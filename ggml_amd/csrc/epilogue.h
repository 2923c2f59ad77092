// epilogue.h — the element-wise tail a MUL_MAT may carry with it: + bias (broadcast over the activation rows), GELU, + residual.
// The gpt-2 graphs (/root/reference/examples/gpt-2/main-backend.cpp:515-521, 595-600, 656-666, 690-698) follow every projection with
// ADD(bias) and then either GELU or ADD(residual); fused, the value never leaves the register it was reduced into.  The arithmetic is the
// SAME sequence of fp32 operations the separate kernels of ops.hip perform (this file is also where they take their GELU from), so the
// fused result is bit-identical to MUL_MAT -> ADD -> (GELU | ADD).
#pragma once
#include "cdna4_common.h"
#include "cdna4_kernels.h"      // struct cdna4_epilogue

__device__ __forceinline__ float gelu_f32(float x) {           // ggml_gelu_f32, ggml-cpu.c:1753-1755
    return 0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x)));
}
// ggml_vec_gelu_f32 (ggml-cpu.c:1759-1774): outside (-10, 10) the closed form's limits, inside a 64K-entry fp16 table indexed by fp16(x)
__device__ __forceinline__ float gelu_lut_f32(float v) {
    if (v <= -10.0f) return 0.0f;
    if (v >= 10.0f) return v;
    const float xh = (float)(half_t)v;
    float r = (float)(half_t)opaque_f32(gelu_f32(xh));     // the table entry is fp16 of the fp32 VALUE (no multiply folded into the conversion)
    if (r == 0.0f) r = __builtin_copysignf(0.0f, xh);          // x * 0 keeps x's sign on the CPU (-0.0 for x <= -5.2): make the zero's sign explicit
    return r;
}
__device__ __forceinline__ float epilogue_apply(const cdna4_epilogue &e, float v, int64_t m, int64_t b) {
    if (e.bias) v = v + e.bias[m];
    if (e.act == 1) v = gelu_lut_f32(v);
    if (e.resid) v = v + e.resid[b * e.resid_row_stride + m];
    return v;
}
