"""-m gpu: the supporting ops and the row converters of include/ggml_cdna4.h called THROUGH THE C-ABI (ctypes, device pointers,
ggml_cdna4_tensor descriptors) and compared with the unmodified reference CPU backend (tests/refops.py -> oracle/_ref) and the
C oracle on identical inputs.  Bars: BIT-EXACT for CPY f32 -> Q8_0 / Q4_0 (bytes: the reference CPU backend's CPY node, whose
dup path calls type_traits_cpu[].from_float — the AVX2 quantize_row_q8_0 of ggml-cpu-quants.c:778-815 and quantize_row_q4_0_ref
of ggml-quants.c:31-66 — plus the quantize_row_q8_0_ref rounding of ggml-cuda's cpy as the second Q8_0 form) and for
dequantize_row of the five formats (to_float, ggml-quants.c:255,349,1280,1482,1690); <= 2e-6 relative L2 for the float ops
(norm, rms_norm, soft_max, rope, gelu, diag_mask_inf, get_rows) whose only freedom is fp32 summation order / libm."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import refutil as R

pytestmark = pytest.mark.gpu
I32 = 26


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    if not R.have_ref():
        pytest.fail("oracle/_ref is missing from the snapshot (run __graft_entry__.build() where /root/reference exists)")
    from ggml_amd import native
    return native.lib()


def _desc(t, type_, ne=None, nb=None):
    """ggml_cdna4_tensor of a torch tensor (contiguous unless ne / nb are given; quantized types: uint8 bytes + explicit ne)"""
    from ggml_amd import native
    d = native.Tensor()
    d.data = t.data_ptr(); d.type = type_; d.reserved = 0
    if ne is None:
        shp = list(reversed(t.shape)) + [1] * (4 - t.dim())
        st = [s * t.element_size() for s in reversed(t.stride())]
        while len(st) < 4:
            st.append(st[-1] * shp[len(st) - 1])
        ne, nb = shp, st
    for i in range(4):
        d.ne[i] = int(ne[i]); d.nb[i] = int(nb[i])
    return d


def _qdesc(t, type_, k, rows):
    rs = R.row_size(type_, k)
    return _desc(t, type_, [k, rows, 1, 1], [R.TYPE_SIZE[type_], rs, rs * rows, rs * rows])


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _ok(L, rc):
    assert rc == 0, L.ggml_cdna4_last_error().decode()


def _data(kind, shape, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, shape).astype(np.float32)
    if kind == "normal":
        return (rng.standard_normal(shape) * 3).astype(np.float32)
    x = (rng.integers(-254, 255, shape) / 2.0).astype(np.float32)        # exact .5 ties for the rounding rules
    return x


# ------------------------------------------------------------------------------------------------ CPY f32 -> quantized: bytes
@pytest.mark.parametrize("kind", ["uniform", "normal", "ties"])
@pytest.mark.parametrize("dst_type,ref_rounding", [(R.Q8_0, 0), (R.Q8_0, 1), (R.Q4_0, 0)])
def test_cpy_f32_to_quantized_is_byte_exact(L, dst_type, ref_rounding, kind):
    """CPY f32 -> Q8_0 / Q4_0 through the C-ABI, byte for byte.  q8_0_ref_rounding = 0 is what the plug-in passes: the CPU backend's
    own from_float (the AVX2 quantize_row_q8_0), i.e. the bytes the REFERENCE's CPY node writes (checked below against the
    compiled reference, ties included); 1 is quantize_row_q8_0_ref, the rounding of ggml-cuda's cpy.  Q4_0 has one form."""
    import refops as O
    rows, k = 37, 1024
    x = _data(kind, (rows, k), 11)
    x[3, 32:64] = 0                                                     # an all-zero block (its scale is -0.0 for Q4_0: 0 / -8)
    x[4, 5] = -7.5; x[4, 9] = 7.5                                        # equal |max| of both signs inside one block
    xd = _dev(x)
    out = torch.zeros(rows * R.row_size(dst_type, k), dtype=torch.uint8, device="cuda")
    _ok(L, L.ggml_cdna4_op_cpy(C.byref(_desc(xd, R.F32)), C.byref(_qdesc(out, dst_type, k, rows)), ref_rounding, _st()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    name = "q4_0_ref" if dst_type == R.Q4_0 else ("q8_0_ref" if ref_rounding else "q8_0_cpu")
    want = np.concatenate([R.o_quantize_row(name, x[i]) for i in range(rows)])
    assert np.array_equal(got, want), "first differing byte %d of %d differing" % (int(np.argmax(got != want)), int((got != want).sum()))
    if not ref_rounding:
        assert np.array_equal(want, O.cpy_quantize(x, dst_type))       # ... and that oracle IS what the reference's CPY node writes


# ------------------------------------------------------------------------------------------------ dequantize_row: bit-exact
@pytest.mark.parametrize("name,t", list(R.QUANT_TYPES.items()))
def test_dequantize_row_is_bit_exact(L, name, t):
    rows, k = 9, 2048
    w = R.random_weights(t, rows, k, seed=int(t) + 1)
    wd = _dev(w)
    y = torch.empty(rows * k, dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_dequantize_row(int(t), wd.data_ptr(), y.data_ptr(), rows * k, _st()))
    torch.cuda.synchronize()
    got = y.cpu().numpy().reshape(rows, k)
    assert np.array_equal(got.view(np.uint32), R.o_dequantize(t, w, k).view(np.uint32))
    assert np.array_equal(got.view(np.uint32), R.r_dequantize(t, w, k).view(np.uint32))


@pytest.mark.parametrize("name,t", list(R.QUANT_TYPES.items()))
def test_cpy_quantized_to_f32_and_get_rows_are_bit_exact(L, name, t):
    import refops as O
    rows, k = 21, 1024
    w = R.random_weights(t, rows, k, seed=int(t) + 3)
    wd = _dev(w)
    y = torch.empty((rows, k), dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_op_cpy(C.byref(_qdesc(wd, t, k, rows)), C.byref(_desc(y, R.F32)), 1, _st()))
    ids = np.array([5, 0, 20, 5, 13], np.int32)
    idd = _dev(ids)
    g = torch.empty((ids.size, k), dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_op_get_rows(C.byref(_qdesc(wd, t, k, rows)), C.byref(_desc(idd, I32)), C.byref(_desc(g, R.F32)), _st()))
    torch.cuda.synchronize()
    want = R.r_dequantize(t, w, k)
    assert np.array_equal(y.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(g.cpu().numpy().view(np.uint32), O.get_rows(t, w, k, rows, ids).view(np.uint32))


# ------------------------------------------------------------------------------------------------ float ops vs the CPU backend
TOL = 2e-6


@pytest.mark.parametrize("rms", [0, 1])
@pytest.mark.parametrize("shape", [(3, 5, 768), (1, 1, 64), (2, 7, 4096)])
def test_norm_and_rms_norm(L, shape, rms):
    import refops as O
    x = _data("normal", shape, 3)
    xd = _dev(x); y = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_norm(C.byref(_desc(xd, R.F32)), C.byref(_desc(y, R.F32)), 1e-5, rms, _st()))
    torch.cuda.synchronize()
    assert R.rel_l2(y.cpu().numpy(), O.norm(x, 1e-5, bool(rms))) < TOL


@pytest.mark.parametrize("mask_dtype,max_bias", [(None, 0.0), (np.float32, 0.0), (np.float16, 0.0), (np.float16, 8.0)])
def test_soft_max(L, mask_dtype, max_bias):
    import refops as O
    n_head, n_q, n_kv = 12, 9, 40
    x = _data("normal", (1, n_head, n_q, n_kv), 4)
    mask = None
    if mask_dtype is not None:
        m = np.zeros((n_q, n_kv), np.float32)
        m[np.triu_indices(n_q, 1, n_kv)] = -np.inf                       # causal
        mask = m.astype(mask_dtype)
    xd = _dev(x); y = torch.empty_like(xd)
    md = _dev(mask) if mask is not None else None
    mt = None if mask is None else C.byref(_desc(md, R.F16 if mask_dtype == np.float16 else R.F32))
    _ok(L, L.ggml_cdna4_op_soft_max(C.byref(_desc(xd, R.F32)), mt, C.byref(_desc(y, R.F32)), 0.125, max_bias, _st()))
    torch.cuda.synchronize()
    assert R.rel_l2(y.cpu().numpy(), O.soft_max(x, mask, 0.125, max_bias)) < TOL


@pytest.mark.parametrize("mode,n_dims,ext", [(0, 64, 0.0), (2, 64, 0.0), (0, 32, 0.0), (2, 128, 1.0)])
def test_rope(L, mode, n_dims, ext):
    import refops as O
    hd = max(n_dims, 64)
    x = _data("uniform", (1, 7, 4, hd), 6)                                # numpy order (ne3, tokens, heads, head_dim)
    pos = np.array([0, 1, 2, 3, 100, 101, 4000], np.int32)
    xd = _dev(x); y = torch.empty_like(xd); pd = _dev(pos)
    args = (n_dims, mode, 4096 if ext else 0, 10000.0, 1.0 if not ext else 0.25, ext, 1.0, 32.0, 1.0)
    _ok(L, L.ggml_cdna4_op_rope(C.byref(_desc(xd, R.F32)), C.byref(_desc(pd, I32)), None, C.byref(_desc(y, R.F32)), *args, _st()))
    torch.cuda.synchronize()
    # sin / cos of the device's libm vs the host's: allow 2e-6 of the vector norm
    assert R.rel_l2(y.cpu().numpy(), O.rope(x, pos, *args)) < 2e-6


@pytest.mark.parametrize("name,op", [("gelu", 0), ("gelu_quick", 1), ("silu", 2)])
def test_unary(L, name, op):
    import refops as O
    x = _data("normal", (5, 333), 8)
    xd = _dev(x); y = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_unary(op, C.byref(_desc(xd, R.F32)), C.byref(_desc(y, R.F32)), _st()))
    torch.cuda.synchronize()
    got, want = y.cpu().numpy(), O.unary(x, name)
    if name == "gelu":
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))     # the CPU's fp16 look-up table semantics, bit for bit
    else:
        assert R.rel_l2(got, want) < TOL


def test_diag_mask_inf(L):
    import refops as O
    x = _data("uniform", (2, 6, 11), 9)
    xd = _dev(x); y = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_diag_mask_inf(C.byref(_desc(xd, R.F32)), C.byref(_desc(y, R.F32)), 3, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy().view(np.uint32), O.diag_mask_inf(x, 3).view(np.uint32))


# ------------------------------------------------------------------------------------------------ fused chains == their node-by-node sequences
def _bin(L, op, a, b, d):
    _ok(L, L.ggml_cdna4_op_binary(op, C.byref(_desc(a, R.F32)), C.byref(_desc(b, R.F32)), C.byref(_desc(d, R.F32)), _st()))


@pytest.mark.parametrize("tail", ["bias", "bias_gelu", "bias_residual"])
@pytest.mark.parametrize("b", [1, 5, 96])
@pytest.mark.parametrize("name,t", [("q4_0", R.Q4_0), ("q4_K", R.Q4_K), ("q8_0", R.Q8_0)])
def test_mul_mat_fused_equals_the_unfused_sequence_bit_for_bit(L, name, t, b, tail):
    """ggml_cdna4_mul_mat_fused against ggml_cdna4_mul_mat -> op_binary(ADD bias) -> op_unary(GELU) | op_binary(ADD residual): the
    one-launch decode form (b = 1), the few-rows GEMV form (b = 5: the tail in the store) and the MFMA GEMM form (b = 96: one
    element-wise launch behind the GEMM) — same bits in all three."""
    m, k = 3072, 768                                                     # gpt-2 117M c_fc
    w = R.random_weights(t, m, k, seed=21)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
    bias = (rng.standard_normal(m) * 3).astype(np.float32)               # spreads the GELU argument over (-10, 10) and beyond
    bias[:4] = [-11.0, 11.0, -6.0, 0.0]
    res = rng.standard_normal((b, m)).astype(np.float32)
    wd, xd, bd, rd = _dev(w), _dev(x), _dev(bias), _dev(res)
    nws = L.ggml_cdna4_mul_mat_workspace_size(int(t), k, b)
    ws = torch.empty(max(nws, 256), dtype=torch.uint8, device="cuda")
    rb = R.row_size(t, k)
    y0 = torch.empty((b, m), dtype=torch.float32, device="cuda"); y1 = torch.empty_like(y0); y2 = torch.empty_like(y0)
    _ok(L, L.ggml_cdna4_mul_mat(int(t), wd.data_ptr(), rb, xd.data_ptr(), k, y0.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, _st()))
    _bin(L, 0, y0, bd, y1)
    if tail == "bias_gelu":
        _ok(L, L.ggml_cdna4_op_unary(0, C.byref(_desc(y1, R.F32)), C.byref(_desc(y2, R.F32)), _st()))
    elif tail == "bias_residual":
        _bin(L, 0, y1, rd, y2)
    else:
        y2 = y1
    yf = torch.full((b, m), 7.0, dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_mul_mat_fused(int(t), wd.data_ptr(), rb, xd.data_ptr(), k, yf.data_ptr(), m, m, k, b, bd.data_ptr(), 1 if tail == "bias_gelu" else 0,
                                      rd.data_ptr() if tail == "bias_residual" else None, m, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    assert np.array_equal(yf.cpu().numpy().view(np.uint32), y2.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("tail", ["bias_gelu", "bias_residual"])
@pytest.mark.parametrize("b", [1, 96, 512])
@pytest.mark.parametrize("name,t", [("q4_K", R.Q4_K), ("q4_0", R.Q4_0)])
def test_mul_mat_fused_against_the_reference_cpu_backend(L, name, t, b, tail):
    """ggml_cdna4_mul_mat_fused against the SAME chain on the unmodified reference CPU backend (tests/refops.py: MUL_MAT -> ADD -> GELU | ADD), not
    only against the unfused HIP sequence.  Q4_K at b = 96 / 512: the tail rides in k_gemm_kq_t64's store — ONE launch per node
    chain; Q4_0: the older GEMM kernels + k_epilogue.  Bar: the MUL_MAT's own bar (GEMV 1e-5 / GEMM 1e-3, relative L2 over the chain's result)."""
    if not os.path.exists(os.path.join(R.REF_DIR, "libggml-cpu.so")):
        pytest.skip("oracle/_ref not built")
    import refops as O
    m, k = 3072, 768
    w = R.random_weights(t, m, k, seed=23)
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
    bias = rng.standard_normal(m).astype(np.float32)
    res = rng.standard_normal((b, m)).astype(np.float32) if tail == "bias_residual" else None
    want = O.mul_mat_tail(t, w, m, k, x, bias, tail == "bias_gelu", res)
    wd, xd, bd = _dev(w), _dev(x), _dev(bias)
    rd = _dev(res) if res is not None else None
    ws = torch.empty(max(L.ggml_cdna4_mul_mat_workspace_size(int(t), k, b), 256), dtype=torch.uint8, device="cuda")
    y = torch.empty((b, m), dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_mul_mat_fused(int(t), wd.data_ptr(), R.row_size(t, k), xd.data_ptr(), k, y.data_ptr(), m, m, k, b, bd.data_ptr(), 1 if tail == "bias_gelu" else 0,
                                      rd.data_ptr() if rd is not None else None, m, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    e = R.rel_l2(y.cpu().numpy(), want)
    assert e < (1e-5 if b <= 8 else 1e-3), e


@pytest.mark.parametrize("tail", ["bias_gelu", "bias_residual", "bias_residual_in_place"])
@pytest.mark.parametrize("t", [R.Q4_K, R.Q5_K])
@pytest.mark.parametrize("m,k,b", [(16384, 1024, 1024), (16384, 512, 2048)])
def test_mul_mat_fused_on_the_large_grid_route(L, m, k, b, tail, t):
    """Q4_K / Q5_K shapes whose 256 x 256 tiles fill the chip take k_gemm_r8 in AUTO (256 and 512 tiles, unsplit):
    the tail rides in ITS store — same bits as MUL_MAT -> ADD -> GELU | ADD on the same route, the product within the GEMM bar of the oracle on sampled
    weight rows, the in-place residual allowed and bit-identical."""
    w = R.random_weights(t, m, k, seed=31)
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
    bias = (rng.standard_normal(m) * 3).astype(np.float32)
    res = rng.standard_normal((b, m)).astype(np.float32)
    wd, xd, bd, rd = _dev(w), _dev(x), _dev(bias), _dev(res)
    ws = torch.empty(max(L.ggml_cdna4_mul_mat_workspace_size(int(t), k, b), 256), dtype=torch.uint8, device="cuda")
    rb = R.row_size(t, k)
    y0 = torch.empty((b, m), dtype=torch.float32, device="cuda"); y1 = torch.empty_like(y0); y2 = torch.empty_like(y0)
    _ok(L, L.ggml_cdna4_mul_mat(int(t), wd.data_ptr(), rb, xd.data_ptr(), k, y0.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, _st()))
    rows = np.random.default_rng(2).choice(m, 40, replace=False)
    wsub = np.concatenate([w[r * rb:(r + 1) * rb] for r in rows])
    assert R.rel_l2(y0.cpu().numpy()[:, rows], R.o_mul_mat(t, wsub, x, len(rows), k)) < 1e-3
    _bin(L, 0, y0, bd, y1)
    if tail == "bias_gelu":
        _ok(L, L.ggml_cdna4_op_unary(0, C.byref(_desc(y1, R.F32)), C.byref(_desc(y2, R.F32)), _st()))
    else:
        _bin(L, 0, y1, rd, y2)
    yf = rd.clone() if tail == "bias_residual_in_place" else torch.full((b, m), 7.0, dtype=torch.float32, device="cuda")
    if tail == "bias_residual_in_place":
        assert L.ggml_cdna4_mul_mat_fused_residual_may_alias(int(t), m, k, b) == 1
    rp = None if tail == "bias_gelu" else (yf.data_ptr() if tail == "bias_residual_in_place" else rd.data_ptr())
    _ok(L, L.ggml_cdna4_mul_mat_fused(int(t), wd.data_ptr(), rb, xd.data_ptr(), k, yf.data_ptr(), m, m, k, b, bd.data_ptr(), 1 if tail == "bias_gelu" else 0, rp, m,
                                      ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    assert np.array_equal(yf.cpu().numpy().view(np.uint32), y2.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("b", [1, 5, 96, 512])
@pytest.mark.parametrize("name,t", [("q4_0", R.Q4_0), ("q4_K", R.Q4_K), ("q6_K", R.Q6_K)])
def test_mul_mat_fused_with_the_residual_in_place(L, name, t, b):
    """residual == Y (ggml_add_inplace(resid, cur), or the graph allocator placing the ADD onto its residual — ADVICE r2, high): where
    ggml_cdna4_mul_mat_fused_residual_may_alias says 1 the in-place call equals the out-of-place one bit for bit; where it says 0 the call is
    refused (never a silently doubled product); a partial overlap is always refused"""
    m, k = 768, 768
    w = R.random_weights(t, m, k, seed=22)
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
    bias = rng.standard_normal(m).astype(np.float32)
    res = rng.standard_normal((b + 1, m)).astype(np.float32)
    wd, xd, bd, rd = _dev(w), _dev(x), _dev(bias), _dev(res)
    ws = torch.empty(max(L.ggml_cdna4_mul_mat_workspace_size(int(t), k, b), 256), dtype=torch.uint8, device="cuda")
    rb = R.row_size(t, k)
    y = torch.empty((b, m), dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_mul_mat_fused(int(t), wd.data_ptr(), rb, xd.data_ptr(), k, y.data_ptr(), m, m, k, b, bd.data_ptr(), 0, rd.data_ptr(), m, ws.data_ptr(), ws.numel(), _st()))
    inplace = rd.clone()
    rc = L.ggml_cdna4_mul_mat_fused(int(t), wd.data_ptr(), rb, xd.data_ptr(), k, inplace.data_ptr(), m, m, k, b, bd.data_ptr(), 0, inplace.data_ptr(), m, ws.data_ptr(), ws.numel(), _st())
    torch.cuda.synchronize()
    if L.ggml_cdna4_mul_mat_fused_residual_may_alias(int(t), m, k, b):
        assert rc == 0, L.ggml_cdna4_last_error()
        assert np.array_equal(inplace[:b].cpu().numpy().view(np.uint32), y.cpu().numpy().view(np.uint32))
    else:
        assert rc != 0 and b"residual" in L.ggml_cdna4_last_error()
        assert np.array_equal(inplace.cpu().numpy(), res)                       # refused before anything was written
    # shifted by one row: partial overlap, refused in every regime
    sh = rd.clone()
    rc = L.ggml_cdna4_mul_mat_fused(int(t), wd.data_ptr(), rb, xd.data_ptr(), k, sh.data_ptr(), m, m, k, b, bd.data_ptr(), 0, sh.data_ptr() + 4 * m, m, ws.data_ptr(), ws.numel(), _st())
    if b > 1:
        assert rc != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("rms", [0, 1])
@pytest.mark.parametrize("with_shift", [False, True])
def test_norm_affine_equals_norm_mul_add(L, rms, with_shift):
    x = _data("normal", (2, 9, 768), 31)
    rng = np.random.default_rng(2)
    g = rng.uniform(0.5, 1.5, 768).astype(np.float32); sh = rng.standard_normal(768).astype(np.float32)
    xd, gd, sd = _dev(x), _dev(g), _dev(sh)
    a = torch.empty_like(xd); b_ = torch.empty_like(xd); c = torch.empty_like(xd); f = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_norm(C.byref(_desc(xd, R.F32)), C.byref(_desc(a, R.F32)), 1e-5, rms, _st()))
    _bin(L, 2, a, gd, b_)
    if with_shift:
        _bin(L, 0, b_, sd, c)
    else:
        c = b_
    _ok(L, L.ggml_cdna4_op_norm_affine(C.byref(_desc(xd, R.F32)), C.byref(_desc(gd, R.F32)), C.byref(_desc(sd, R.F32)) if with_shift else None,
                                       C.byref(_desc(f, R.F32)), 1e-5, rms, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy().view(np.uint32), c.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("n_past,n_q", [(0, 7), (40, 1), (13, 5)])
def test_soft_max_ext_equals_scale_mask_soft_max(L, n_past, n_q):
    n_head, n_kv = 12, n_past + n_q
    x = _data("normal", (1, n_head, n_q, n_kv), 44)
    xd = _dev(x); a = torch.empty_like(xd); b_ = torch.empty_like(xd); c = torch.empty_like(xd); f = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_scale(C.byref(_desc(xd, R.F32)), C.byref(_desc(a, R.F32)), 0.125, _st()))
    _ok(L, L.ggml_cdna4_op_diag_mask_inf(C.byref(_desc(a, R.F32)), C.byref(_desc(b_, R.F32)), n_past, _st()))
    _ok(L, L.ggml_cdna4_op_soft_max(C.byref(_desc(b_, R.F32)), None, C.byref(_desc(c, R.F32)), 1.0, 0.0, _st()))
    _ok(L, L.ggml_cdna4_op_soft_max_ext(C.byref(_desc(xd, R.F32)), None, C.byref(_desc(f, R.F32)), 1.0, 0.0, 1, 0.125, n_past, _st()))
    # ... and in place, as the gpt-2 graph runs it (ggml_scale_inplace / diag_mask_inf_inplace / soft_max_inplace on one buffer)
    g = xd.clone()
    _ok(L, L.ggml_cdna4_op_soft_max_ext(C.byref(_desc(g, R.F32)), None, C.byref(_desc(g, R.F32)), 1.0, 0.0, 1, 0.125, n_past, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy().view(np.uint32), c.cpu().numpy().view(np.uint32))
    assert np.array_equal(g.cpu().numpy().view(np.uint32), c.cpu().numpy().view(np.uint32))
