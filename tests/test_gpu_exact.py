"""-m gpu: GGML_CDNA4_EXACT's kernels (ggml_amd/csrc/exact.hip) against the UNMODIFIED reference CPU backend (tests/refops.py -> oracle/_ref, the x86-64-v3
build of oracle/ref.mk) — BIT FOR BIT: MUL_MAT on Q4_0 / Q8_0 weights (ggml_vec_dot_q4_0_q8_0 / _q8_0_q8_0 in their AVX2 lane order) and on Q4_K / Q5_K / Q6_K weights
(ggml_vec_dot_q4_K_q8_K / _q5_K_q8_K / _q6_K_q8_K: lane sums, Q4_K's four-lane acc_m, Q5_K's scalar summs), RMS_NORM, SILU, MUL_MAT F32 x F32
(ggml_vec_dot_f32 incl. gcc's leftover loop), NORM (sequential double sums) and SOFT_MAX (ggml_v_expf per chunk of eight, glibc's expf on the tail).
These are the three ops of a gpt-2 graph that differ from the CPU backend by fp32 summation order in the default mode (VERDICT r3 "weak 1")."""
import ctypes as C

import numpy as np
import pytest
import torch

import refutil as R
from test_gpu_cabi_ops import L, _desc, _dev, _ok, _st      # noqa: F401  (the fixture + descriptor helpers)

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("wtype", [R.Q4_0, R.Q8_0])
@pytest.mark.parametrize("m,k,b", [(64, 768, 1), (48, 768, 8), (33, 3072, 5), (16, 256, 17), (96, 64, 9)])
def test_mul_mat_exact_is_the_cpu_backends_bits(L, wtype, m, k, b):
    import refops
    rng = np.random.default_rng(m * 7 + k + b)
    w = R.random_weights(wtype, m, k, seed=m + k)
    x = (rng.standard_normal((b, k)) * np.exp(rng.uniform(-2, 2, (b, 1)))).astype(np.float32)
    want = refops.mul_mat(wtype, w, m, k, x)
    wd, xd = _dev(w), _dev(x)
    y = torch.empty((b, m), dtype=torch.float32, device="cuda")
    nws = L.ggml_cdna4_mul_mat_exact_workspace_size(int(wtype), k, b)
    assert nws > 0 and L.ggml_cdna4_mul_mat_exact_supported(int(wtype), k) == 1
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    _ok(L, L.ggml_cdna4_mul_mat_exact(int(wtype), wd.data_ptr(), R.row_size(wtype, k), xd.data_ptr(), k, y.data_ptr(), m, m, k, b, ws.data_ptr(), nws, _st()))
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), (np.abs(got - want).max(), int((_bits(got) != _bits(want)).sum()))


@pytest.mark.parametrize("wtype", [R.Q4_K, R.Q5_K, R.Q6_K])
@pytest.mark.parametrize("m,k,b", [(64, 1024, 1), (48, 768, 8), (33, 2816, 5), (16, 256, 17), (96, 4096, 9)])
def test_mul_mat_exact_k_quants_are_the_cpu_backends_bits(L, wtype, m, k, b):
    """rows of mixed magnitude (the mins term and the lane sums differ by orders), one all-zero row (d = 0), one row with a single spike (iscale = -127 / max picks the sign)"""
    import refops
    rng = np.random.default_rng(m * 7 + k + b + int(wtype))
    w = R.random_weights(wtype, m, k, seed=m + k)
    x = (rng.standard_normal((b, k)) * np.exp(rng.uniform(-2, 2, (b, 1)))).astype(np.float32)
    if b >= 5:
        x[1, :] = 0.0
        x[2, :] = 1e-3 * x[2, :]; x[2, k // 3] = -7.5
    want = refops.mul_mat(wtype, w, m, k, x)
    wd, xd = _dev(w), _dev(x)
    y = torch.empty((b, m), dtype=torch.float32, device="cuda")
    nws = L.ggml_cdna4_mul_mat_exact_workspace_size(int(wtype), k, b)
    assert nws > 0 and L.ggml_cdna4_mul_mat_exact_supported(int(wtype), k) == 1 and L.ggml_cdna4_mul_mat_exact_supported(int(wtype), k + 32) == 0
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    _ok(L, L.ggml_cdna4_mul_mat_exact(int(wtype), wd.data_ptr(), R.row_size(wtype, k), xd.data_ptr(), k, y.data_ptr(), m, m, k, b, ws.data_ptr(), nws, _st()))
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), (np.abs(got - want).max(), int((_bits(got) != _bits(want)).sum()))


@pytest.mark.parametrize("rows,n", [(5, 1024), (64, 768), (3, 1000), (2, 33)])
def test_rms_norm_exact_is_the_cpu_backends_bits(L, rows, n):
    import refops
    rng = np.random.default_rng(rows + 3 * n)
    x = (rng.standard_normal((rows, n)) * np.exp(rng.uniform(-3, 3, (rows, 1)))).astype(np.float32)
    want = refops.norm(x, 1e-5, True)
    xd = _dev(x)
    y = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_rms_norm_exact(C.byref(_desc(xd, R.F32)), C.byref(_desc(y, R.F32)), 1e-5, _st()))
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), (np.abs(got - want).max(), int((_bits(got) != _bits(want)).sum()))


@pytest.mark.parametrize("rows,nc", [(3, 2816), (5, 100), (7, 13), (4, 8), (2, 7), (9, 1)])
def test_silu_exact_is_the_cpu_backends_bits(L, rows, nc):
    import refops
    rng = np.random.default_rng(rows * 17 + nc)
    x = (rng.standard_normal((rows, nc)) * 6).astype(np.float32)
    x[0, :] = np.linspace(-110, 95, nc, dtype=np.float32)               # both ends of ggml_v_expf's range (flush to zero / to infinity)
    want = refops.unary(x, "silu")
    xd = _dev(x)
    y = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_silu_exact(C.byref(_desc(xd, R.F32)), C.byref(_desc(y, R.F32)), _st()))
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), (np.abs(got - want).max(), int((_bits(got) != _bits(want)).sum()))


@pytest.mark.parametrize("m,n,k,n2a,n2", [(5, 7, 64, 1, 1), (9, 3, 8, 2, 2), (4, 6, 13, 1, 3), (7, 2, 45, 2, 4), (3, 5, 100, 1, 1), (6, 4, 1, 1, 1), (2, 3, 39, 1, 2), (8, 8, 160, 1, 1)])
def test_mul_mat_f32_exact_is_the_cpu_backends_bits(L, m, n, k, n2a, n2):
    import refops
    rng = np.random.default_rng(m + 10 * n + 100 * k)
    a = rng.standard_normal((1, n2a, m, k)).astype(np.float32)
    b = rng.standard_normal((1, n2, n, k)).astype(np.float32)
    want = refops.mul_mat_f32_batched(a, b)
    ad, bd = _dev(a), _dev(b)
    y = torch.empty((1, n2, n, m), dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_op_mul_mat_f_exact(C.byref(_desc(ad, R.F32)), C.byref(_desc(bd, R.F32)), C.byref(_desc(y, R.F32)), _st()))
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), (np.abs(got - want).max(), int((_bits(got) != _bits(want)).sum()))


@pytest.mark.parametrize("rows,n", [(5, 768), (64, 768), (3, 1000), (2, 33)])
def test_norm_exact_is_the_cpu_backends_bits(L, rows, n):
    import refops
    rng = np.random.default_rng(rows + n)
    x = (rng.standard_normal((rows, n)) * 3 + rng.uniform(-1, 1, (rows, 1))).astype(np.float32)
    want = refops.norm(x, 1e-5, False)
    xd = _dev(x)
    y = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_norm_exact(C.byref(_desc(xd, R.F32)), C.byref(_desc(y, R.F32)), 1e-5, _st()))
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), (np.abs(got - want).max(), int((_bits(got) != _bits(want)).sum()))


@pytest.mark.parametrize("rows,nc", [(12, 8), (12, 9), (24, 14), (7, 64), (5, 100), (3, 1), (4, 23)])
@pytest.mark.parametrize("masked", [False, True])
def test_soft_max_exact_is_the_cpu_backends_bits(L, rows, nc, masked):
    import refops
    rng = np.random.default_rng(rows * 31 + nc)
    x = (rng.standard_normal((rows, nc)) * 4).astype(np.float32)
    x[0, :] = np.linspace(-120, 0, nc, dtype=np.float32)              # deep tail of the exponential (denormal results, the |n| > 126 branch)
    if masked:                                                          # what DIAG_MASK_INF leaves: -inf right of a diagonal
        for r_ in range(rows):
            x[r_, (r_ % nc) + 1:] = -np.inf
    want = refops.soft_max(x, None, 1.0, 0.0)
    xd = _dev(x)
    y = torch.empty_like(xd)
    _ok(L, L.ggml_cdna4_op_soft_max_exact(C.byref(_desc(xd, R.F32)), C.byref(_desc(y, R.F32)), 1.0, _st()))
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), (np.abs(got - want).max(), int((_bits(got) != _bits(want)).sum()))
