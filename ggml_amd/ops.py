"""Host-side mirror of the reference's operator interface for the hot path, over the C-ABI.

Names and argument meaning follow ggml (include/ggml.h): `mul_mat(a, b)` is ggml_mul_mat (src/ggml.c:2694-2709):
a = weights [K, M] (ggml ne order) block-quantized, b = activations [K, B] f32, result [M, B] f32 — in torch's
row-major shapes: a.bytes (M, row_size), b (B, K), result (B, M).  torch is used only for device memory and the
current HIP stream; all arithmetic happens in libcdna4_kernels.so.  No CPU path exists here.
"""
import torch

from . import native
from .gtypes import GGMLType, row_size, type_size, QUANT_WEIGHT_TYPES

PATH_AUTO, PATH_GEMV, PATH_GEMM = 0, 1, 2


class QTensor:
    """a block-quantized 2-D weight tensor living in HBM: ne = (K, M) like ggml, bytes (M, row_size(type, K))"""

    def __init__(self, type, K, M, data):
        self.type = GGMLType(type)
        if self.type not in QUANT_WEIGHT_TYPES:
            raise ValueError("unsupported weight type %r" % (type,))
        self.K, self.M = int(K), int(M)
        self.row_bytes = row_size(self.type, self.K)
        if data.dtype != torch.uint8 or data.numel() != self.M * self.row_bytes:
            raise ValueError("weight bytes: expected %d uint8, got %s %d" % (self.M * self.row_bytes, data.dtype, data.numel()))
        if not data.is_cuda:
            raise native.NativeError("QTensor data must live on the GPU (there is no CPU path)")
        self.data = data.contiguous().view(self.M, self.row_bytes)

    @classmethod
    def from_host_bytes(cls, type, K, M, host_bytes, device="cuda"):
        """host_bytes: numpy uint8 array / bytes holding M rows of ggml blocks (e.g. from a GGUF payload)"""
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(np.frombuffer(host_bytes, dtype=np.uint8) if isinstance(host_bytes, (bytes, bytearray)) else host_bytes).reshape(-1).copy())
        return cls(type, K, M, t.to(device))

    def rows(self, lo, hi):
        """row slice [lo, hi) — the unit of the row-split (output-feature) sharding across GPUs"""
        return QTensor(self.type, self.K, hi - lo, self.data[lo:hi].reshape(-1))


_ws = {}


def _workspace(dev, nbytes):
    buf = _ws.get(dev)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws[dev] = buf
    return buf


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _need_gpu(t, what):
    if not t.is_cuda:
        raise native.NativeError("%s must be a GPU tensor (there is no CPU path)" % what)


def _same_device(dev, **tensors):
    """the library keys its per-device state (split-K scratch, CU count) on the CURRENT device: every operand must live on
    the device the call is issued for, and the call is issued with that device current"""
    for name, t in tensors.items():
        if t is not None and t.device != dev:
            raise ValueError("%s lives on %s, expected %s" % (name, t.device, dev))


def _check_out(out, B, M, dev):
    if out.dtype != torch.float32 or out.dim() != 2 or tuple(out.shape) != (B, M) or out.stride(1) != 1 or out.device != dev:
        raise ValueError("out must be float32 (%d, %d) with contiguous rows on %s" % (B, M, dev))


def mul_mat(a: QTensor, b: torch.Tensor, out=None, path=PATH_AUTO, gemm_variant=0, splitk=0):
    """ggml_mul_mat(a, b) for quantized a, f32 b of shape (B, K).  Returns f32 (B, M)."""
    L = native.lib()
    _need_gpu(b, "b")
    if b.dtype != torch.float32 or b.dim() != 2 or b.shape[1] != a.K or b.stride(1) != 1:
        raise ValueError("b must be float32 (B, K=%d) with contiguous rows" % a.K)   # ggml_can_mul_mat, src/ggml.c:2686
    B = b.shape[0]
    dev = b.device
    _same_device(dev, a=a.data)
    if out is None:
        out = torch.empty((B, a.M), dtype=torch.float32, device=dev)
    else:
        _check_out(out, B, a.M, dev)
    nws = L.ggml_cdna4_mul_mat_workspace_size(int(a.type), a.K, max(B, 1))
    ws = _workspace(dev, nws)
    with torch.cuda.device(dev):
        native.check(L.ggml_cdna4_mul_mat(int(a.type), a.data.data_ptr(), a.row_bytes, b.data_ptr(), b.stride(0),
                                          out.data_ptr(), out.stride(0), a.M, a.K, B, ws.data_ptr(), ws.numel(),
                                          path, gemm_variant, splitk, _stream(dev)))
    return out


class PreparedAct:
    """activations quantized once (Q8_K / Q8_0 like the CPU backend) and kept resident; reusable across
    every weight matrix of the same type family and K (e.g. the q/k/v projections of one layer)."""

    def __init__(self, wtype, b: torch.Tensor, path=PATH_AUTO, M_hint=4096):
        L = native.lib()
        _need_gpu(b, "b")
        self.type, self.B, self.K, self.path = GGMLType(wtype), b.shape[0], b.shape[1], path
        n = L.ggml_cdna4_mul_mat_workspace_size(int(self.type), self.K, max(self.B, 1))
        self.ws = torch.empty(n, dtype=torch.uint8, device=b.device)
        with torch.cuda.device(b.device):
            native.check(L.ggml_cdna4_prepare_act(int(self.type), b.data_ptr(), b.stride(0), self.K, self.B,
                                                  self.ws.data_ptr(), self.ws.numel(), path, _stream(b.device)))


def mul_mat_prepared(a: QTensor, act: PreparedAct, out=None, path=None, gemm_variant=0, splitk=0):
    L = native.lib()
    if act.K != a.K:
        raise ValueError("K mismatch")
    dev = act.ws.device
    _same_device(dev, a=a.data)
    if out is None:
        out = torch.empty((act.B, a.M), dtype=torch.float32, device=dev)
    else:
        _check_out(out, act.B, a.M, dev)
    with torch.cuda.device(dev):
        native.check(L.ggml_cdna4_mul_mat_prepared(int(a.type), a.data.data_ptr(), a.row_bytes, out.data_ptr(), out.stride(0),
                                                   a.M, a.K, act.B, act.ws.data_ptr(), act.ws.numel(),
                                                   act.path if path is None else path, gemm_variant, splitk, _stream(dev)))
    return out


def mul_mat_id(as_: "list[QTensor] | QTensor", b: torch.Tensor, ids: torch.Tensor, n_expert=None):
    """ggml_mul_mat_id (src/ggml.c:2735-2759): as_ = stacked experts (n_expert*M rows), b (n_tok, n_b, K) f32,
    ids (n_tok, n_used) int32 -> (n_tok, n_used, M) f32."""
    L = native.lib()
    a = as_
    _need_gpu(b, "b")
    n_tok, n_b, K = b.shape
    n_used = ids.shape[1]
    M = a.M // n_expert
    out = torch.empty((n_tok, n_used, M), dtype=torch.float32, device=b.device)
    nws = L.ggml_cdna4_mul_mat_id_workspace_size(int(a.type), K, n_expert, n_used, n_b, n_tok)
    ws = _workspace(b.device, nws)
    b = b.contiguous()
    ids = ids.to(torch.int32).contiguous()
    _same_device(b.device, a=a.data, ids=ids)
    with torch.cuda.device(b.device):
      native.check(L.ggml_cdna4_mul_mat_id(int(a.type), a.data.data_ptr(), a.row_bytes, M * a.row_bytes,
                                         b.data_ptr(), b.stride(1), b.stride(0), ids.data_ptr(), ids.stride(0),
                                         out.data_ptr(), out.stride(1), out.stride(0), M, K, n_expert, n_used, n_b, n_tok,
                                         ws.data_ptr(), ws.numel(), _stream(b.device)))
    return out


def quantize_row_q8_K(x: torch.Tensor, want_f16=False):
    """from_float of GGML_TYPE_Q8_K (quantize_row_q8_K_ref, src/ggml-quants.c:2479-2516) over the rows of x (B, K).
    Returns (qs int8 (B,K), d f32 (B,K/256), bsums int16 (B,K/16)[, xh fp16 (B,K) pair-interleaved])."""
    L = native.lib()
    _need_gpu(x, "x")
    B, K = x.shape
    qs = torch.empty((B, K), dtype=torch.int8, device=x.device)
    d = torch.empty((B, K // 256), dtype=torch.float32, device=x.device)
    bs = torch.empty((B, K // 16), dtype=torch.int16, device=x.device)
    xh = torch.empty((B, (K + 127) // 128 * 128), dtype=torch.float16, device=x.device) if want_f16 else None
    native.check(L.ggml_cdna4_quantize_q8_K(x.data_ptr(), x.stride(0), K, B, qs.data_ptr(), d.data_ptr(), bs.data_ptr(),
                                            xh.data_ptr() if want_f16 else None, _stream(x.device)))
    return (qs, d, bs, xh) if want_f16 else (qs, d, bs)


def quantize_row_q8_0(x: torch.Tensor, ref_rounding=False, want_f16=False):
    """from_float of GGML_TYPE_Q8_0: AVX2 body (src/ggml-cpu/ggml-cpu-quants.c:778-815) or `_ref`
    (src/ggml-quants.c:194-217).  Returns (qs int8 (B,K), d f32 (B,K/32) holding the fp16-rounded scale[, xh])."""
    L = native.lib()
    _need_gpu(x, "x")
    B, K = x.shape
    qs = torch.empty((B, K), dtype=torch.int8, device=x.device)
    d = torch.empty((B, K // 32), dtype=torch.float32, device=x.device)
    xh = torch.empty((B, (K + 127) // 128 * 128), dtype=torch.float16, device=x.device) if want_f16 else None
    native.check(L.ggml_cdna4_quantize_q8_0(x.data_ptr(), x.stride(0), K, B, qs.data_ptr(), d.data_ptr(),
                                            xh.data_ptr() if want_f16 else None, 1 if ref_rounding else 0, _stream(x.device)))
    return (qs, d, xh) if want_f16 else (qs, d)


def quantize_row_q8_1(x: torch.Tensor, want_f16=False):
    """from_float of GGML_TYPE_Q8_1 as the CPU backend runs it for the activations of Q4_1 / Q5_1 weights (AVX2 body of quantize_row_q8_1,
    src/ggml-cpu/ggml-cpu-quants.c:1076-1119).  Returns (qs int8 (B,K), d f32 (B,K/32) = the fp16-rounded scale, s f32 (B,K/32) =
    fp16(d * sum q) with d still in fp32[, xh])."""
    L = native.lib()
    _need_gpu(x, "x")
    B, K = x.shape
    qs = torch.empty((B, K), dtype=torch.int8, device=x.device)
    d = torch.empty((B, K // 32), dtype=torch.float32, device=x.device)
    s = torch.empty((B, K // 32), dtype=torch.float32, device=x.device)
    xh = torch.empty((B, (K + 127) // 128 * 128), dtype=torch.float16, device=x.device) if want_f16 else None
    native.check(L.ggml_cdna4_quantize_q8_1(x.data_ptr(), x.stride(0), K, B, qs.data_ptr(), d.data_ptr(), s.data_ptr(),
                                            xh.data_ptr() if want_f16 else None, _stream(x.device)))
    return (qs, d, s, xh) if want_f16 else (qs, d, s)


def convert_weights(a: QTensor) -> QTensor:
    """exact re-encoding of a Q5_0 / Q3_K matrix as Q8_0 / Q6_K (ggml_cdna4_convert_weights): same dequantized values bit for bit, and
    the MFMA prefill GEMM of the target format.  mul_mat does this per call for more than 8 activation rows; converting once trades
    HBM (34/22 resp. 210/110 of the bytes) for the conversion pass."""
    L = native.lib()
    tgt = L.ggml_cdna4_convert_weights_target(int(a.type))
    if tgt < 0:
        raise ValueError("%s has no exact target format" % a.type.name)
    dev = a.data.device
    out = torch.empty(L.ggml_cdna4_convert_weights_size(int(a.type), a.M, a.K), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        native.check(L.ggml_cdna4_convert_weights(int(a.type), a.data.data_ptr(), a.row_bytes, a.M, a.K, out.data_ptr(), _stream(dev)))
    return QTensor(tgt, a.K, a.M, out)


def _tensor_desc(t: torch.Tensor, type_):
    """ggml_cdna4_tensor of a 4-D torch tensor: ggml's ne / nb are torch's shape / strides reversed (strides in bytes)"""
    d = native.Tensor()
    d.data = t.data_ptr(); d.type = int(type_); d.reserved = 0
    for i in range(4):
        d.ne[i] = int(t.shape[3 - i]); d.nb[i] = int(t.stride(3 - i)) * t.element_size()
    return d


def _qrows_desc(t: torch.Tensor, type_, D):
    """ggml_cdna4_tensor of block-quantized rows held as uint8 (batch, n_head_kv, n_kv, row_size(type, D)): ne = (D, n_kv, n_head_kv, batch)"""
    if t.dtype != torch.uint8 or t.dim() != 4 or t.stride(3) != 1 or t.shape[3] != row_size(type_, D):
        raise ValueError("quantized k / v: uint8 (batch, n_head_kv, n_kv, %d) with contiguous rows" % row_size(type_, D))
    d = native.Tensor()
    d.data = t.data_ptr(); d.type = int(type_); d.reserved = 0
    d.ne[0] = int(D); d.nb[0] = type_size(type_)
    for i in range(1, 4):
        d.ne[i] = int(t.shape[3 - i]); d.nb[i] = int(t.stride(3 - i))
    return d


def flash_attn_ext(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask=None, scale=1.0, max_bias=0.0, logit_softcap=0.0, kv_type=None):
    """ggml_flash_attn_ext (include/ggml.h:1758-1767; CPU: ggml-cpu.c:10805-11016).  q f32 (batch, n_head, n_q, D) — any strides with
    contiguous rows, e.g. a permuted view; k, v fp16 or bf16 (batch_kv, n_head_kv, n_kv, D); mask fp16 (>= n_q, n_kv) or None.
    kv_type (Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0): k, v are block-quantized rows as uint8 (batch_kv, n_head_kv, n_kv, row_size(kv_type, D)).
    Returns f32 (batch, n_q, n_head, D) like ggml's result (ne = D, n_head, n_q, batch)."""
    import ctypes as C
    L = native.lib()
    for name, t in (("q", q), ("k", k), ("v", v)):
        _need_gpu(t, name)
        if t.dim() != 4 or t.stride(3) != 1:
            raise ValueError("%s must be 4-D with contiguous rows" % name)
    kv16 = {torch.float16: GGMLType.F16, torch.bfloat16: GGMLType.BF16}
    if q.dtype != torch.float32 or (kv_type is None and (k.dtype not in kv16 or v.dtype not in kv16)):
        raise ValueError("q must be float32, k and v float16 / bfloat16 (or uint8 block rows with kv_type)")
    dev = q.device
    _same_device(dev, k=k, v=v, mask=mask)
    if mask is not None and (mask.dtype != torch.float16 or mask.dim() != 2 or not mask.is_contiguous()):
        raise ValueError("mask must be a contiguous float16 (>= n_q, n_kv) matrix")
    B3, H, N, D = q.shape
    out = torch.empty((B3, N, H, D), dtype=torch.float32, device=dev)
    dm = _tensor_desc(mask.view(1, 1, *mask.shape), GGMLType.F16) if mask is not None else None
    dq, dd = _tensor_desc(q, GGMLType.F32), _tensor_desc(out, GGMLType.F32)
    dk, dv = (_tensor_desc(k, kv16[k.dtype]), _tensor_desc(v, kv16[v.dtype])) if kv_type is None else (_qrows_desc(k, kv_type, D), _qrows_desc(v, kv_type, D))
    with torch.cuda.device(dev):
        native.check(L.ggml_cdna4_op_flash_attn_ext(C.byref(dq), C.byref(dk), C.byref(dv), C.byref(dm) if dm is not None else None, C.byref(dd),
                                                    float(scale), float(max_bias), float(logit_softcap), _stream(dev)))
    return out
