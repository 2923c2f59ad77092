// ggml_cdna4_ops.cpp — supporting-op dispatch of the plug-in: ggml_tensor -> plain-pointer descriptors
// (ggml_cdna4_tensor) -> kernels of libcdna4_kernels.so.  supports_* mirrors exactly what the kernels accept;
// everything else is declined so the scheduler leaves it on the CPU backend.
#include "ggml_cdna4_ops.h"
#include <stdlib.h>
#include "ggml_cdna4.h"
#include <cstdio>
#include <cstring>

static ggml_cdna4_tensor desc(const ggml_tensor * t) {
    ggml_cdna4_tensor d;
    d.data = t->data; d.type = (int32_t)t->type; d.reserved = 0;
    for (int i = 0; i < 4; i++) { d.ne[i] = t->ne[i]; d.nb[i] = (int64_t)t->nb[i]; }
    return d;
}
static bool f32(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32; }
// quantized sources of GET_ROWS / CPY -> F32 (to_float of ops.hip)
static bool qsrc(ggml_type t) { return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K ||
                                       t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K || t == GGML_TYPE_IQ4_NL || t == GGML_TYPE_IQ4_XS; }
static bool ff16(ggml_type t) { return t == GGML_TYPE_F32 || t == GGML_TYPE_F16; }

static int unary_id(const ggml_tensor * op) {
    switch (ggml_get_unary_op(op)) {
        case GGML_UNARY_OP_GELU: return GGML_CDNA4_GELU;
        case GGML_UNARY_OP_GELU_QUICK: return GGML_CDNA4_GELU_QUICK;
        case GGML_UNARY_OP_SILU: return GGML_CDNA4_SILU;
        case GGML_UNARY_OP_RELU: return GGML_CDNA4_RELU;
        case GGML_UNARY_OP_TANH: return GGML_CDNA4_TANH;
        default: return -1;
    }
}

bool cdna4_ops_supports_matmul(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    return ff16(a->type) && f32(b) && f32(op) && a->nb[0] == ggml_type_size(a->type) && b->nb[0] == sizeof(float) && ggml_is_contiguous(op);
}

bool cdna4_ops_supports_tensor(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    switch (op->op) {
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV:
            return f32(a) && f32(b) && f32(op) && ggml_can_repeat(b, a);
        case GGML_OP_SCALE:
            return f32(a) && f32(op) && ggml_is_contiguous(a) && ggml_is_contiguous(op);
        case GGML_OP_UNARY:
            return f32(a) && f32(op) && ggml_is_contiguous(a) && ggml_is_contiguous(op) && unary_id(op) >= 0;
        case GGML_OP_NORM: case GGML_OP_RMS_NORM:
            return f32(a) && f32(op) && a->nb[0] == sizeof(float) && op->nb[0] == sizeof(float);
        case GGML_OP_SOFT_MAX:
            return f32(a) && f32(op) && ggml_is_contiguous(a) && ggml_is_contiguous(op) &&
                   (!b || ((b->type == GGML_TYPE_F32 || b->type == GGML_TYPE_F16) && ggml_is_contiguous(b)));
        case GGML_OP_DIAG_MASK_INF:
            return f32(a) && f32(op) && ggml_is_contiguous(a) && ggml_is_contiguous(op);
        case GGML_OP_GET_ROWS:
            return (ff16(a->type) || qsrc(a->type)) && b->type == GGML_TYPE_I32 && f32(op) && a->nb[0] == ggml_type_size(a->type);
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            const ggml_type ta = a->type, td = op->type;
            if (ff16(ta) && ff16(td)) return true;
            if (qsrc(ta) && td == GGML_TYPE_F32) return a->nb[0] == ggml_type_size(ta);
            if (ta == GGML_TYPE_F32 && (td == GGML_TYPE_Q8_0 || td == GGML_TYPE_Q4_0 || td == GGML_TYPE_Q4_1 || td == GGML_TYPE_Q5_0 || td == GGML_TYPE_Q5_1))
                return a->nb[0] == sizeof(float) && a->ne[0] % 32 == 0 && ggml_is_contiguous(op);
            return false;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            // q F32, k / v F16 with 16-byte aligned rows, BF16 or block-quantized, mask F16 contiguous or absent, head sizes up to 256 (64 / 128 / 256 directly,
            // others zero-padded: fattn.hip)
            const ggml_tensor * k = op->src[1], * v = op->src[2], * m = op->src[3];
            // (a quantized k / v — Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0 — is copied out as fp16 first: any strides, contiguous rows)
            if (!f32(a) || !f32(op) || !k || !v) return false;
            if (!ggml_cdna4_op_flash_attn_ext_supported(a->ne[0], (int)k->type) || !ggml_cdna4_op_flash_attn_ext_supported(a->ne[0], (int)v->type)) return false;
            if (k->ne[0] != a->ne[0] || v->ne[0] != a->ne[0]) return false;
            if (a->nb[0] != sizeof(float) || k->nb[0] != ggml_type_size(k->type) || v->nb[0] != ggml_type_size(v->type) || !ggml_is_contiguous(op)) return false;
            for (int i = 1; i < 4; i++) if ((k->type == GGML_TYPE_F16 && k->nb[i] % 16) || (v->type == GGML_TYPE_F16 && v->nb[i] % 16) || a->nb[i] % 4) return false;
            if (m && (m->type != GGML_TYPE_F16 || !ggml_is_contiguous(m))) return false;
            return a->ne[2] <= 65535 && a->ne[3] <= 65535;
        }
        case GGML_OP_ROPE: {
            const int mode = ((const int32_t *)op->op_params)[2];
            return f32(a) && f32(op) && b->type == GGML_TYPE_I32 && (mode & ~2) == 0 && (!op->src[2] || op->src[2]->type == GGML_TYPE_F32);
        }
        default:
            return false;
    }
}

// GGML_CDNA4_EXACT=1: the ops whose fp32 summation order differs from the CPU backend's by default run in the CPU's own order (exact.hip); a whole gpt-2
// graph then reproduces the CPU backend's logits bit for bit (tests/test_gpu_gpt2.py).  Verification mode: slower.
bool cdna4_exact_mode() { static const bool on = getenv("GGML_CDNA4_EXACT") && atoi(getenv("GGML_CDNA4_EXACT")) != 0; return on; }

enum ggml_status cdna4_ops_compute(void * ctx, ggml_tensor * node) {
    void * stream = cdna4_backend_stream(ctx);
    const ggml_tensor * a = node->src[0], * b = node->src[1];
    ggml_cdna4_tensor da = a ? desc(a) : ggml_cdna4_tensor{}, db = b ? desc(b) : ggml_cdna4_tensor{}, dd = desc(node);
    int rc = -1;
    switch (node->op) {
        case GGML_OP_ADD: rc = ggml_cdna4_op_binary(GGML_CDNA4_ADD, &da, &db, &dd, stream); break;
        case GGML_OP_SUB: rc = ggml_cdna4_op_binary(GGML_CDNA4_SUB, &da, &db, &dd, stream); break;
        case GGML_OP_MUL: rc = ggml_cdna4_op_binary(GGML_CDNA4_MUL, &da, &db, &dd, stream); break;
        case GGML_OP_DIV: rc = ggml_cdna4_op_binary(GGML_CDNA4_DIV, &da, &db, &dd, stream); break;
        case GGML_OP_SCALE: { float s; memcpy(&s, node->op_params, sizeof(float)); rc = ggml_cdna4_op_scale(&da, &dd, s, stream); break; }
        case GGML_OP_UNARY:
            if (cdna4_exact_mode() && ggml_get_unary_op(node) == GGML_UNARY_OP_SILU && a->type == GGML_TYPE_F32 && ggml_is_contiguous(a) && ggml_is_contiguous(node)) rc = ggml_cdna4_op_silu_exact(&da, &dd, stream);
            else rc = ggml_cdna4_op_unary(unary_id(node), &da, &dd, stream);
            break;
        case GGML_OP_NORM: case GGML_OP_RMS_NORM: {
            float eps; memcpy(&eps, node->op_params, sizeof(float));
            if (cdna4_exact_mode() && node->op == GGML_OP_NORM) rc = ggml_cdna4_op_norm_exact(&da, &dd, eps, stream);
            else if (cdna4_exact_mode() && a->type == GGML_TYPE_F32) rc = ggml_cdna4_op_rms_norm_exact(&da, &dd, eps, stream);
            else rc = ggml_cdna4_op_norm(&da, &dd, eps, node->op == GGML_OP_RMS_NORM, stream);
            break;
        }
        case GGML_OP_SOFT_MAX: {
            float scale, max_bias; memcpy(&scale, (const float *)node->op_params + 0, 4); memcpy(&max_bias, (const float *)node->op_params + 1, 4);
            if (cdna4_exact_mode() && !b && max_bias == 0.0f && ggml_is_contiguous(a) && ggml_is_contiguous(node)) rc = ggml_cdna4_op_soft_max_exact(&da, &dd, scale, stream);
            else rc = ggml_cdna4_op_soft_max(&da, b ? &db : nullptr, &dd, scale, max_bias, stream);
            break;
        }
        case GGML_OP_DIAG_MASK_INF: rc = ggml_cdna4_op_diag_mask_inf(&da, &dd, ((const int32_t *)node->op_params)[0], stream); break;
        case GGML_OP_GET_ROWS: rc = ggml_cdna4_op_get_rows(&da, &db, &dd, stream); break;
        // F32 -> Q8_0 with the CPU backend's own from_float (type_traits_cpu[Q8_0].from_float = the SIMD quantize_row_q8_0, which is what
        // ggml_compute_forward_dup_f32 calls: ggml-cpu.c:3230-3260) — measured: byte-identical to the reference's CPY, whereas
        // quantize_row_q8_0_ref (ggml-cuda's rounding) differs from it on exact .5 ties
        // F32 -> Q8_0: the rounding of the CPU backend's from_float on the oracle's x86 hosts (ggml-cpu.c:2965 -> quantize_row_q8_0,
        // AVX2 form: id = 127 / amax, nearest-even), which is also what the activation quantizer uses; quantize_row_q8_0_ref and
        // ggml-cuda/cpy.cu:61 use id = 1 / d with roundf instead (C-ABI callers get that with q8_0_ref_rounding = 1)
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: rc = ggml_cdna4_op_cpy(&da, &dd, /*q8_0_ref_rounding=*/0, stream); break;
        case GGML_OP_MUL_MAT:
            if (cdna4_exact_mode() && a->type == GGML_TYPE_F32) rc = ggml_cdna4_op_mul_mat_f_exact(&da, &db, &dd, stream);
            else rc = ggml_cdna4_op_mul_mat_f(&da, &db, &dd, stream);
            break;
        case GGML_OP_FLASH_ATTN_EXT: {
            float scale, max_bias, softcap;
            memcpy(&scale, (const float *)node->op_params + 0, 4); memcpy(&max_bias, (const float *)node->op_params + 1, 4); memcpy(&softcap, (const float *)node->op_params + 2, 4);
            ggml_cdna4_tensor dv = desc(node->src[2]), dm = node->src[3] ? desc(node->src[3]) : ggml_cdna4_tensor{};
            rc = ggml_cdna4_op_flash_attn_ext(&da, &db, &dv, node->src[3] ? &dm : nullptr, &dd, scale, max_bias, softcap, stream); break;
        }
        case GGML_OP_ROPE: {
            const int32_t * ip = (const int32_t *)node->op_params;
            float fb, fs, ef, af, bf, bs;
            memcpy(&fb, ip + 5, 4); memcpy(&fs, ip + 6, 4); memcpy(&ef, ip + 7, 4); memcpy(&af, ip + 8, 4); memcpy(&bf, ip + 9, 4); memcpy(&bs, ip + 10, 4);
            ggml_cdna4_tensor dc = node->src[2] ? desc(node->src[2]) : ggml_cdna4_tensor{};
            rc = ggml_cdna4_op_rope(&da, &db, node->src[2] ? &dc : nullptr, &dd, ip[1], ip[2], ip[4], fb, fs, ef, af, bf, bs, stream); break;
        }
        default: break;
    }
    if (rc != 0) { fprintf(stderr, "ggml-cdna4: %s: %s\n", ggml_op_name(node->op), ggml_cdna4_last_error()); return GGML_STATUS_FAILED; }
    return GGML_STATUS_SUCCESS;
}
