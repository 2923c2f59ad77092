"""ctypes access to the CPU oracle (oracle/libggml_oracle.so) and, when present, to the unmodified
reference compiled by oracle/ref.mk (oracle/_ref/libggml-*.so).  Test infrastructure only."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libggml_oracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

# ggml type ids (include/ggml.h:351-390)
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 8, 12, 13, 14, 15
Q4_1, Q5_0, Q5_1, Q8_1, Q2_K, Q3_K = 3, 6, 7, 9, 10, 11        # GEMV units + an MFMA prefill path through exact re-encodings
IQ4_NL, IQ4_XS = 20, 23                                         # the same
QUANT_TYPES = {"q4_0": Q4_0, "q8_0": Q8_0, "q4_K": Q4_K, "q5_K": Q5_K, "q6_K": Q6_K}          # the formats of the HIP path
ORACLE_ONLY_TYPES = {"q4_1": Q4_1, "q5_0": Q5_0, "q5_1": Q5_1, "q2_K": Q2_K, "q3_K": Q3_K, "iq4_nl": IQ4_NL, "iq4_xs": IQ4_XS}
TYPE_SIZE = {F32: 4, F16: 2, Q4_0: 18, Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_1: 36, Q2_K: 84, Q3_K: 110, IQ4_NL: 18, IQ4_XS: 136}
BLCK = {F32: 1, F16: 1, Q4_0: 32, Q8_0: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_1: 32, Q2_K: 256, Q3_K: 256, IQ4_NL: 32, IQ4_XS: 256}


def row_size(t, k):
    return k // BLCK[t] * TYPE_SIZE[t]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/libggml_oracle.so missing: run `make -C oracle`")
        o = C.CDLL(ORACLE_SO)
        o.oracle_vec_dot.restype = C.c_float
        o.oracle_fp16_to_fp32.restype = C.c_float
        o.oracle_fp16_to_fp32.argtypes = [C.c_uint16]
        o.oracle_fp32_to_fp16.restype = C.c_uint16
        o.oracle_fp32_to_fp16.argtypes = [C.c_float]
        _oracle = o
    return _oracle


def o_dequantize(t, wbytes, k):
    n = wbytes.size // row_size(t, k)
    y = np.empty((n, k), np.float32)
    for r in range(n):
        oracle().oracle_dequantize_row(C.c_int(t), _p(wbytes[r * row_size(t, k):]), _p(y[r]), C.c_int64(k))
    return y


def act_type(wtype):
    """type_traits_cpu[wtype].vec_dot_type (src/ggml-cpu/ggml-cpu.c:253-418)"""
    return Q8_0 if wtype in (Q4_0, Q8_0, Q5_0, IQ4_NL) else (Q8_1 if wtype in (Q4_1, Q5_1) else Q8_K)


def o_quantize_act(wtype, x):
    """activation quantization exactly as the CPU backend does for a weight of type wtype"""
    x = np.ascontiguousarray(x, np.float32)
    at = act_type(wtype)
    b, k = x.shape
    out = np.zeros((b, row_size(at, k)), np.uint8)
    for r in range(b):
        oracle().oracle_quantize_act(C.c_int(wtype), _p(x[r]), _p(out[r]), C.c_int64(k))
    return out


def o_quantize_row(name, x):
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    t = {"q4_0_ref": Q4_0, "q8_0_ref": Q8_0, "q8_0_cpu": Q8_0, "q8_K": Q8_K, "q4_1_ref": Q4_1, "q5_0_ref": Q5_0, "q5_1_ref": Q5_1}[name]
    out = np.zeros(row_size(t, x.size), np.uint8)
    getattr(oracle(), "oracle_quantize_row_" + name)(_p(x), _p(out), C.c_int64(x.size))
    return out


def o_mul_mat(t, w, x, m, k, exact=False):
    x = np.ascontiguousarray(x, np.float32)
    b = x.shape[0]
    y = np.empty((b, m), np.float32)
    fn = oracle().oracle_mul_mat_exact if exact else oracle().oracle_mul_mat
    fn(C.c_int(t), _p(w), _p(x), _p(y), C.c_int64(m), C.c_int64(k), C.c_int64(b))
    return y


def o_mul_mat_id(t, w, x, ids, m, k, n_expert):
    x = np.ascontiguousarray(x, np.float32)      # [n_tok][n_b][K]
    ids = np.ascontiguousarray(ids, np.int32)    # [n_tok][n_used]
    n_tok, n_b, _ = x.shape
    n_used = ids.shape[1]
    y = np.empty((n_tok, n_used, m), np.float32)
    oracle().oracle_mul_mat_id(C.c_int(t), _p(w), _p(x), _p(ids), _p(y), C.c_int64(m), C.c_int64(k),
                               C.c_int64(n_expert), C.c_int64(n_used), C.c_int64(n_b), C.c_int64(n_tok))
    return y


def o_flash_attn_ext(q, k, v, mask, scale, max_bias=0.0, logit_softcap=0.0):
    """oracle_flash_attn_ext_f16: q f32 (n_batch, n_head, n_q, D); k, v fp16 (n_batch_kv, n_head_kv, n_kv, D); mask fp16 (>= n_q, n_kv) or None
    -> f32 (n_batch, n_q, n_head, D)  (ggml_compute_forward_flash_attn_ext_f16, ggml-cpu.c:10805-11016)"""
    nb, nh, nq, D = q.shape
    nbk, nhk, nkv, _ = k.shape
    q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float16); v = np.ascontiguousarray(v, np.float16)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.float16)
        assert mask.shape[1] == nkv and mask.shape[0] >= nq
    y = np.empty((nb, nq, nh, D), np.float32)
    oracle().oracle_flash_attn_ext_f16(_p(q), _p(k), _p(v), _p(mask) if mask is not None else None, _p(y), C.c_int64(D), C.c_int64(nq), C.c_int64(nh),
                                       C.c_int64(nb), C.c_int64(nkv), C.c_int64(nhk), C.c_int64(nbk), C.c_float(scale), C.c_float(max_bias), C.c_float(logit_softcap))
    return y


def alibi_slopes(n_head, max_bias):
    """ggml-cpu.c:10880-10902"""
    if max_bias <= 0:
        return np.ones(n_head)
    n2 = 1 << int(np.floor(np.log2(n_head)))
    m0, m1 = 2.0 ** (-max_bias / n2), 2.0 ** (-(max_bias / 2.0) / n2)
    return np.array([m0 ** (h + 1) if h < n2 else m1 ** (2 * (h - n2) + 1) for h in range(n_head)])


def exact_flash_attn_ext(q, k, v, mask, scale, max_bias=0.0, logit_softcap=0.0):
    """the same operator in float64 on the fp16-rounded Q, K, V (what both the CPU and the GPU path start from): the yardstick that shows
    which side's rounding a difference comes from"""
    nb, nh, nq, D = q.shape
    nbk, nhk, nkv, _ = k.shape
    qd = q.astype(np.float16).astype(np.float64); kd = k.astype(np.float64); vd = v.astype(np.float64)
    sl = alibi_slopes(nh, max_bias)
    y = np.empty((nb, nq, nh, D), np.float64)
    for b in range(nb):
        for h in range(nh):
            kk, vv = kd[b // (nb // nbk), h // (nh // nhk)], vd[b // (nb // nbk), h // (nh // nhk)]
            s = qd[b, h] @ kk.T * (scale / logit_softcap if logit_softcap else scale)
            if logit_softcap:
                s = logit_softcap * np.tanh(s)
            if mask is not None:
                s = s + sl[h] * mask[:nq].astype(np.float64)
            s = s - s.max(axis=1, keepdims=True)
            pr = np.exp(s)
            y[b, :, h] = (pr / pr.sum(axis=1, keepdims=True)) @ vv
    return y


# ---------------------------------------------------------------------------------------------
# the unmodified reference (oracle/_ref), when built
_ref = None


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libggml-cpu.so"))


def ref():
    global _ref
    if _ref is None:
        base = C.CDLL(os.path.join(REF_DIR, "libggml-base.so"), mode=C.RTLD_GLOBAL)
        cpu = C.CDLL(os.path.join(REF_DIR, "libggml-cpu.so"), mode=C.RTLD_GLOBAL)
        base.ggml_quantize_chunk.restype = C.c_size_t
        base.ggml_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]

        class InitParams(C.Structure):      # struct ggml_init_params, include/ggml.h:624-629
            _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]
        base.ggml_init.restype = C.c_void_p
        base.ggml_init.argtypes = [InitParams]
        base._InitParams = InitParams
        base.ggml_free.argtypes = [C.c_void_p]
        # first ggml_init fills the fp16->fp32 table the base library's GGML_FP16_TO_FP32 reads (src/ggml.c:1390-1420)
        base.ggml_free(base.ggml_init(InitParams(1 << 20, None, False)))
        cpu.ggml_cpu_init()
        _ref = (base, cpu)
    return _ref


def r_quantize(t, x):
    """ggml_quantize_chunk(type, src, dst, start=0, nrows, n_per_row, imatrix=NULL) — src/ggml.c:6410"""
    x = np.ascontiguousarray(x, np.float32)
    rows, k = x.shape
    out = np.zeros(rows * row_size(t, k), np.uint8)
    base, _ = ref()
    base.ggml_quantize_init(C.c_int(t))
    n = base.ggml_quantize_chunk(t, _p(x), _p(out), 0, rows, k, None)
    assert n == out.size
    return out


_DEQ = {Q4_0: "dequantize_row_q4_0", Q8_0: "dequantize_row_q8_0", Q4_K: "dequantize_row_q4_K",
        Q5_K: "dequantize_row_q5_K", Q6_K: "dequantize_row_q6_K", Q8_K: "dequantize_row_q8_K",
        Q4_1: "dequantize_row_q4_1", Q5_0: "dequantize_row_q5_0", Q5_1: "dequantize_row_q5_1", Q2_K: "dequantize_row_q2_K", Q3_K: "dequantize_row_q3_K",
        IQ4_NL: "dequantize_row_iq4_nl", IQ4_XS: "dequantize_row_iq4_xs"}


def r_dequantize(t, wbytes, k):
    base, _ = ref()
    n = wbytes.size // row_size(t, k)
    y = np.empty((n, k), np.float32)
    getattr(base, _DEQ[t])(_p(wbytes), _p(y), C.c_int64(n * k))
    return y


def r_quantize_act(wtype, x):
    """what ggml_compute_forward_mul_mat does to src1: type_traits_cpu[vec_dot_type].from_float"""
    base, cpu = ref()
    x = np.ascontiguousarray(x, np.float32)
    b, k = x.shape
    if act_type(wtype) == Q8_0:
        out = np.zeros((b, row_size(Q8_0, k)), np.uint8)
        for r in range(b):
            cpu.quantize_row_q8_0(_p(x[r]), _p(out[r]), C.c_int64(k))
    elif act_type(wtype) == Q8_1:
        out = np.zeros((b, row_size(Q8_1, k)), np.uint8)
        for r in range(b):
            cpu.quantize_row_q8_1(_p(x[r]), _p(out[r]), C.c_int64(k))
    else:
        out = np.zeros((b, row_size(Q8_K, k)), np.uint8)
        for r in range(b):
            cpu.quantize_row_q8_K(_p(x[r]), _p(out[r]), C.c_int64(k))
    return out


_VD = {Q4_0: "ggml_vec_dot_q4_0_q8_0", Q8_0: "ggml_vec_dot_q8_0_q8_0", Q4_K: "ggml_vec_dot_q4_K_q8_K",
       Q5_K: "ggml_vec_dot_q5_K_q8_K", Q6_K: "ggml_vec_dot_q6_K_q8_K", Q4_1: "ggml_vec_dot_q4_1_q8_1", Q5_0: "ggml_vec_dot_q5_0_q8_0",
       Q5_1: "ggml_vec_dot_q5_1_q8_1", Q2_K: "ggml_vec_dot_q2_K_q8_K", Q3_K: "ggml_vec_dot_q3_K_q8_K",
       IQ4_NL: "ggml_vec_dot_iq4_nl_q8_0", IQ4_XS: "ggml_vec_dot_iq4_xs_q8_K"}


def r_mul_mat(t, w, x, m, k):
    """reference MUL_MAT semantics through the reference's own from_float + vec_dot symbols"""
    _, cpu = ref()
    act = r_quantize_act(t, x)
    b = act.shape[0]
    y = np.empty((b, m), np.float32)
    fn = getattr(cpu, _VD[t])
    s = C.c_float()
    rs = row_size(t, k)
    wp = w.ctypes.data
    for bi in range(b):
        ap = act[bi].ctypes.data
        for mi in range(m):
            fn(C.c_int(k), C.byref(s), C.c_size_t(0), C.c_void_p(wp + mi * rs), C.c_size_t(0), C.c_void_p(ap), C.c_size_t(0), C.c_int(1))
            y[bi, mi] = s.value
    return y


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def random_weights(t, m, k, seed):
    """quantized weight bytes.  With the reference present: uniform(-1,1) through ggml_quantize_chunk
    (what tests/test-backend-ops.cpp:37-126 does).  Without it: random but VALID block bytes."""
    rng = np.random.default_rng(seed)
    if have_ref():
        return r_quantize(t, rng.uniform(-1, 1, (m, k)).astype(np.float32))
    return random_block_bytes(t, m, k, rng)


def random_block_bytes(t, m, k, rng):
    nb = m * k // BLCK[t]
    raw = rng.integers(0, 256, (nb, TYPE_SIZE[t]), dtype=np.uint8)
    def f16(lo, hi, n):
        return rng.uniform(lo, hi, n).astype(np.float16).view(np.uint8).reshape(n, 2)
    if t in (Q4_0, Q8_0, IQ4_NL):
        raw[:, 0:2] = f16(-0.2, 0.2, nb)
    elif t == IQ4_XS:
        raw[:, 0:2] = f16(-0.01, 0.01, nb)
    elif t in (Q4_K, Q5_K):
        raw[:, 0:2] = f16(0.001, 0.02, nb); raw[:, 2:4] = f16(0.001, 0.02, nb)
    elif t == Q6_K:
        raw[:, 208:210] = f16(-0.01, 0.01, nb)
    return raw.reshape(-1)
