"""-m gpu parity tests of the rows SURVEY.md 8(f) ranks after the five headline formats: the remaining block formats (rank 4).
Same bars as test_gpu_parity.py: rel-L2 <= 1e-5 for the int8-dot GEMV units, <= 1e-3 for the fp16-MFMA GEMM, bit-exact for to_float and
for the weight re-encodings.  (This file sorts last on purpose: what it covers is newer than the five-format path.)"""
import numpy as np
import pytest
import torch
import refutil as R

pytestmark = pytest.mark.gpu

GEMV_TYPES = [("q5_0", R.Q5_0), ("q2_K", R.Q2_K), ("q3_K", R.Q3_K)]           # int8-dot units
GEMM_TYPES = [("q5_0", R.Q5_0, R.Q8_0), ("q3_K", R.Q3_K, R.Q6_K)]              # prefill through the exact re-encoding
TOFLOAT_TYPES = [("q4_1", R.Q4_1), ("q5_0", R.Q5_0), ("q5_1", R.Q5_1), ("q2_K", R.Q2_K), ("q3_K", R.Q3_K)]
TOL_GEMV, TOL_GEMM = 1e-5, 1e-3


@pytest.fixture(scope="module")
def gu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    import gpu_util
    from ggml_amd import native
    native.lib()
    return gpu_util


def _x(seed, b, k):
    return np.random.default_rng(seed).uniform(-1, 1, (b, k)).astype(np.float32)


@pytest.mark.parametrize("name,t", GEMV_TYPES)
@pytest.mark.parametrize("m,k,b", [(16, 256, 1), (48, 1024, 8), (33, 2048, 5), (256, 4096, 2), (20, 512, 19)])
def test_more_formats_gemv_parity(gu, name, t, m, k, b):
    """vec_dot_q5_0_q8_0 / q2_K_q8_K / q3_K_q8_K (src/ggml-cpu/ggml-cpu-quants.c) through the GEMV units, 1..19 activation rows"""
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=m + k)
    x = _x(b + k, b, k)
    y = ops.mul_mat(gu.qtensor(t, w, m, k), gu.to_dev(x), path=ops.PATH_GEMV).cpu().numpy()
    e = R.rel_l2(y, R.o_mul_mat(t, w, x, m, k)); gu.report(test="more_formats_gemv", type=name, m=m, k=k, b=b, rel_l2=e)
    assert np.isfinite(y).all() and e < TOL_GEMV


@pytest.mark.parametrize("name,t,tgt", GEMM_TYPES)
@pytest.mark.parametrize("m,k", [(9, 512), (130, 2048), (257, 1024)])
def test_weight_reencoding_is_exact(gu, name, t, tgt, m, k):
    """ggml_cdna4_convert_weights: dequantize_row(target bytes) == dequantize_row(source bytes) bit for bit (oracle on both sides)"""
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=3 * m + k)
    c = ops.convert_weights(gu.qtensor(t, w, m, k))
    assert int(c.type) == tgt and c.data.numel() == m * R.row_size(tgt, k)
    cw = c.data.cpu().numpy().reshape(-1)
    assert np.array_equal(R.o_dequantize(tgt, cw, k).view(np.uint32), R.o_dequantize(t, w, k).view(np.uint32))


@pytest.mark.parametrize("name,t,tgt", GEMM_TYPES)
@pytest.mark.parametrize("m,k,b", [(16, 256, 9), (200, 1024, 100), (130, 768, 33), (512, 2048, 128), (4096, 4096, 512)])
def test_more_formats_prefill_gemm(gu, name, t, tgt, m, k, b):
    """above 8 activation rows Q5_0 / Q3_K run the MFMA GEMM of Q8_0 / Q6_K on the re-encoded weights: within the GEMM bar of the oracle's
    MUL_MAT for the SOURCE format, and bit-identical to the target format's GEMM on weights converted up front"""
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=5 * m + k)
    x = _x(b * 7 + k, b, k)
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y = ops.mul_mat(a, xd).cpu().numpy()
    assert np.isfinite(y).all()
    if m <= 512:
        e = R.rel_l2(y, R.o_mul_mat(t, w, x, m, k)); gu.report(test="more_formats_gemm", type=name, m=m, k=k, b=b, rel_l2=e)
        assert e < TOL_GEMM
    else:                                                       # full size: a 64-row sample of weight rows against the oracle
        rows = np.random.default_rng(0).choice(m, 64, replace=False); rs = R.row_size(t, k)
        wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
        e = R.rel_l2(y[:, rows], R.o_mul_mat(t, wsub, x, 64, k)); gu.report(test="more_formats_gemm", type=name, m=m, k=k, b=b, rel_l2=e)
        assert e < TOL_GEMM
    y2 = ops.mul_mat(ops.convert_weights(a), xd).cpu().numpy()
    assert np.array_equal(y, y2)
    # a second matrix of the same shape must not see the first one's re-encoded weights (scratch is rebuilt per call)
    w3 = R.random_weights(t, m, k, seed=991)
    a3 = gu.qtensor(t, w3, m, k)
    assert np.array_equal(ops.mul_mat(a3, xd).cpu().numpy(), ops.mul_mat(ops.convert_weights(a3), xd).cpu().numpy())


@pytest.mark.parametrize("name,t", TOFLOAT_TYPES)
def test_more_formats_to_float_is_bit_exact(gu, name, t):
    """ggml_cdna4_dequantize_row for Q4_1 / Q5_0 / Q5_1 / Q2_K / Q3_K against the oracle and the compiled reference (dequantize_row_*)"""
    from ggml_amd import native
    L = native.lib()
    rows, k = 9, 2048
    w = R.random_weights(t, rows, k, seed=int(t) + 1)
    wd = gu.to_dev(w)
    y = torch.empty(rows * k, dtype=torch.float32, device="cuda")
    native.check(L.ggml_cdna4_dequantize_row(int(t), wd.data_ptr(), y.data_ptr(), rows * k, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = y.cpu().numpy().reshape(rows, k)
    assert np.array_equal(got.view(np.uint32), R.o_dequantize(t, w, k).view(np.uint32))
    if R.have_ref():
        assert np.array_equal(got.view(np.uint32), R.r_dequantize(t, w, k).view(np.uint32))
