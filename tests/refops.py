"""Single ggml ops evaluated by the UNMODIFIED reference CPU backend (oracle/_ref/libggml-base.so + libggml-cpu.so, built by
oracle/ref.mk from /root/reference) through ctypes: a context, the op constructor of include/ggml.h, ggml_graph_compute_with_ctx
(include/ggml-cpu.h:71).  Test infrastructure: the checker for the C-ABI's supporting ops (tests/test_gpu_cabi_ops.py)."""
import ctypes as C

import numpy as np

import refutil as R

_NP = {R.F32: np.float32, R.F16: np.float16, 26: np.int32}     # 26 = GGML_TYPE_I32


class Ref:
    """one context per evaluation (ggml_init / ggml_free, include/ggml.h:624-700)"""

    def __init__(self, mem=256 << 20):
        self.base, self.cpu = R.ref()
        b = self.base
        if not getattr(b, "_refops_ready", False):
            vp, i64, i32, f = C.c_void_p, C.c_int64, C.c_int, C.c_float
            b.ggml_new_tensor_4d.restype = vp; b.ggml_new_tensor_4d.argtypes = [vp, i32, i64, i64, i64, i64]
            b.ggml_get_data.restype = vp; b.ggml_get_data.argtypes = [vp]
            b.ggml_nbytes.restype = C.c_size_t; b.ggml_nbytes.argtypes = [vp]
            for name, args in (("ggml_norm", [vp, vp, f]), ("ggml_rms_norm", [vp, vp, f]), ("ggml_soft_max_ext", [vp, vp, vp, f, f]),
                               ("ggml_rope_ext", [vp, vp, vp, vp, i32, i32, i32, f, f, f, f, f, f]), ("ggml_cpy", [vp, vp, vp]),
                               ("ggml_gelu", [vp, vp]), ("ggml_gelu_quick", [vp, vp]), ("ggml_silu", [vp, vp]), ("ggml_get_rows", [vp, vp, vp]),
                               ("ggml_diag_mask_inf", [vp, vp, i32]), ("ggml_add", [vp, vp, vp]), ("ggml_mul", [vp, vp, vp]), ("ggml_scale", [vp, vp, f]),
                               ("ggml_mul_mat", [vp, vp, vp]), ("ggml_cont", [vp, vp]), ("ggml_permute", [vp, vp, i32, i32, i32, i32]),
                               ("ggml_flash_attn_ext", [vp, vp, vp, vp, vp, f, f, f])):
                fn = getattr(b, name); fn.restype = vp; fn.argtypes = args
            b.ggml_new_graph.restype = vp; b.ggml_new_graph.argtypes = [vp]
            b.ggml_build_forward_expand.argtypes = [vp, vp]
            self.cpu.ggml_graph_compute_with_ctx.restype = i32; self.cpu.ggml_graph_compute_with_ctx.argtypes = [vp, vp, i32]
            b._refops_ready = True

        # tests/ggufref.py binds ggml_init on the same library object with its own (identical) structure class: use whichever is bound
        params_t = (b.ggml_init.argtypes or [b._InitParams])[0]
        self.ctx = b.ggml_init(params_t(mem, None, False))
        assert self.ctx

    def close(self):
        if self.ctx:
            self.base.ggml_free(self.ctx); self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def tensor(self, type_, ne, data=None):
        """new tensor with ggml shape ne (ne[0] fastest); data: numpy array (typed for F32/F16/I32, raw bytes for quantized types)"""
        ne = list(ne) + [1] * (4 - len(ne))
        t = self.base.ggml_new_tensor_4d(self.ctx, type_, *ne)
        assert t
        if data is not None:
            raw = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
            n = self.base.ggml_nbytes(t)
            assert raw.size == n, (raw.size, n)
            C.memmove(self.base.ggml_get_data(t), raw.ctypes.data, n)
        return t

    def compute(self, t, n_threads=4):
        g = self.base.ggml_new_graph(self.ctx)
        self.base.ggml_build_forward_expand(g, t)
        assert self.cpu.ggml_graph_compute_with_ctx(self.ctx, g, n_threads) == 0
        return t

    def read(self, t, type_, shape=None):
        """the tensor's bytes as numpy (typed for F32/F16/I32, uint8 otherwise); shape in numpy order (slowest first)"""
        n = self.base.ggml_nbytes(t)
        buf = (C.c_uint8 * n).from_address(self.base.ggml_get_data(t))
        a = np.frombuffer(buf, np.uint8).copy()
        if type_ in _NP:
            a = a.view(_NP[type_])
        return a.reshape(shape) if shape is not None else a


def _np_shape(ne):
    return tuple(reversed([int(x) for x in ne]))


def norm(x, eps, rms):
    """ggml_norm / ggml_rms_norm over ne[0]; x: numpy f32, last axis = ne[0]"""
    ne = list(reversed(x.shape))
    with Ref() as r:
        a = r.tensor(R.F32, ne, x.astype(np.float32))
        t = (r.base.ggml_rms_norm if rms else r.base.ggml_norm)(r.ctx, a, C.c_float(eps))
        return r.read(r.compute(t), R.F32, x.shape)


def soft_max(x, mask, scale, max_bias):
    ne = list(reversed(x.shape))
    with Ref() as r:
        a = r.tensor(R.F32, ne, x.astype(np.float32))
        m = None
        if mask is not None:
            m = r.tensor(R.F16 if mask.dtype == np.float16 else R.F32, list(reversed(mask.shape)), mask)
        t = r.base.ggml_soft_max_ext(r.ctx, a, m, C.c_float(scale), C.c_float(max_bias))
        return r.read(r.compute(t), R.F32, x.shape)


def rope(x, pos, n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow, freq_factors=None):
    """x: numpy (ne3, ne2 = n_tokens, ne1 = n_head, ne0 = head_dim)-shaped f32 in numpy order; pos: int32 (ne2,)"""
    ne = list(reversed(x.shape))
    with Ref() as r:
        a = r.tensor(R.F32, ne, x.astype(np.float32))
        p = r.tensor(26, [pos.size], pos.astype(np.int32))
        ff = r.tensor(R.F32, [freq_factors.size], freq_factors.astype(np.float32)) if freq_factors is not None else None
        t = r.base.ggml_rope_ext(r.ctx, a, p, ff, n_dims, mode, n_ctx_orig, C.c_float(freq_base), C.c_float(freq_scale), C.c_float(ext_factor),
                                 C.c_float(attn_factor), C.c_float(beta_fast), C.c_float(beta_slow))
        return r.read(r.compute(t), R.F32, x.shape)


def cpy_quantize(x, dst_type):
    """ggml_cpy(f32 -> dst_type): the bytes the CPU backend's dup path writes (ggml_compute_forward_dup, ggml-cpu.c:2860-4050)"""
    ne = list(reversed(x.shape))
    with Ref() as r:
        a = r.tensor(R.F32, ne, x.astype(np.float32))
        d = r.tensor(dst_type, ne)
        t = r.base.ggml_cpy(r.ctx, a, d)
        return r.read(r.compute(t), dst_type)


def unary(x, name):
    with Ref() as r:
        a = r.tensor(R.F32, list(reversed(x.shape)), x.astype(np.float32))
        t = getattr(r.base, "ggml_" + name)(r.ctx, a)
        return r.read(r.compute(t), R.F32, x.shape)


def get_rows(src_type, src_bytes_or_f32, k, nrows, ids):
    """rows `ids` (int32 1-D) of a [k, nrows] tensor of src_type, dequantized to f32"""
    with Ref() as r:
        a = r.tensor(src_type, [k, nrows], src_bytes_or_f32)
        i = r.tensor(26, [ids.size], ids.astype(np.int32))
        t = r.base.ggml_get_rows(r.ctx, a, i)
        return r.read(r.compute(t), R.F32, (ids.size, k))


def diag_mask_inf(x, n_past):
    with Ref() as r:
        a = r.tensor(R.F32, list(reversed(x.shape)), x.astype(np.float32))
        t = r.base.ggml_diag_mask_inf(r.ctx, a, n_past)
        return r.read(r.compute(t), R.F32, x.shape)


def flash_attn_ext(q, k, v, mask, scale, max_bias=0.0, logit_softcap=0.0, n_threads=4):
    """ggml_flash_attn_ext (include/ggml.h:1758-1767) on the reference CPU backend.  q f32 (n_batch, n_head, n_q, D); k, v fp16
    (n_batch_kv, n_head_kv, n_kv, D); mask fp16 (GGML_PAD(n_q, GGML_KQ_MASK_PAD = 64), n_kv) or None -> f32 (n_batch, n_q, n_head, D)"""
    nb, nh, nq, D = q.shape
    with Ref(mem=1 << 30) as r:
        tq = r.tensor(R.F32, list(reversed(q.shape)), q.astype(np.float32))
        tk = r.tensor(R.F16, list(reversed(k.shape)), k.astype(np.float16))
        tv = r.tensor(R.F16, list(reversed(v.shape)), v.astype(np.float16))
        tm = r.tensor(R.F16, list(reversed(mask.shape)), mask.astype(np.float16)) if mask is not None else None
        t = r.base.ggml_flash_attn_ext(r.ctx, tq, tk, tv, tm, C.c_float(scale), C.c_float(max_bias), C.c_float(logit_softcap))
        r.compute(t, n_threads)
        return r.read(t, R.F32, (nb, nq, nh, D))


def mul_mat_tail(wtype, w_bytes, m, k, x, bias, gelu, resid, n_threads=8):
    """the chain the gpt-2 graphs put behind every projection (examples/gpt-2/main-backend.cpp:515-521, 656-666), evaluated node by node on the
    reference CPU backend: ggml_mul_mat(W, x) -> ggml_add(bias) -> ggml_gelu | ggml_add(residual).  x (B, K) f32 -> (B, M) f32"""
    b = x.shape[0]
    with Ref(mem=(1 << 28) + w_bytes.size + 16 * b * (m + k)) as r:
        tw = r.tensor(wtype, [k, m], w_bytes)
        tx = r.tensor(R.F32, [k, b], x.astype(np.float32))
        t = r.base.ggml_mul_mat(r.ctx, tw, tx)
        if bias is not None:
            t = r.base.ggml_add(r.ctx, t, r.tensor(R.F32, [m], bias.astype(np.float32)))
        if gelu:
            t = r.base.ggml_gelu(r.ctx, t)
        if resid is not None:
            t = r.base.ggml_add(r.ctx, t, r.tensor(R.F32, [m, b], resid.astype(np.float32)))
        return r.read(r.compute(t, n_threads), R.F32, (b, m))


def mul_mat(wtype, w_bytes, m, k, x, n_threads=8):
    """ggml_mul_mat(W [k, m] of wtype, x [k, b] f32) on the reference CPU backend -> (b, m) f32 (w_bytes: raw block bytes, or f32 / f16 values)"""
    b = x.shape[0]
    with Ref(mem=(1 << 28) + int(np.asarray(w_bytes).nbytes) + 16 * b * (m + k)) as r:
        tw = r.tensor(wtype, [k, m], w_bytes)
        tx = r.tensor(R.F32, [k, b], x.astype(np.float32))
        return r.read(r.compute(r.base.ggml_mul_mat(r.ctx, tw, tx), n_threads), R.F32, (b, m))


def mul_mat_f32_batched(a, b, n_threads=4):
    """ggml_mul_mat of two f32 tensors with batch dims: a (n3a, n2a, m, k), b (n3, n2, n, k) numpy (slowest first) -> (n3, n2, n, m)"""
    with Ref() as r:
        ta = r.tensor(R.F32, list(reversed(a.shape)), a.astype(np.float32))
        tb = r.tensor(R.F32, list(reversed(b.shape)), b.astype(np.float32))
        t = r.compute(r.base.ggml_mul_mat(r.ctx, ta, tb), n_threads)
        return r.read(t, R.F32, (b.shape[0], b.shape[1], b.shape[2], a.shape[2]))
