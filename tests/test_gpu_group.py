"""-m gpu: ggml_cdna4_mul_mat_group through the C-ABI — n one-row MUL_MATs of one activation row in ONE launch (k_gemv_q_fused_grp) against the n separate calls
(ggml_cdna4_mul_mat / _mul_mat_fused with B = 1), BIT FOR BIT, and against the oracle; shapes without a grouped form answer -2 and launch nothing."""
import ctypes as C

import numpy as np
import pytest
import torch

import refutil as R
from test_gpu_cabi_ops import L, _dev, _ok, _st      # noqa: F401  (the fixture + helpers)

pytestmark = pytest.mark.gpu


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


@pytest.mark.parametrize("t", [R.Q4_K, R.Q5_K, R.Q6_K, R.Q4_0, R.Q8_0])
@pytest.mark.parametrize("ms,k,biased", [((4096, 1024, 1024), 4096, (False, False, True)), ((1408, 1408), 2048, (False, False)), ((50, 33, 17, 260), 512, (True, False, True, False)),
                                         ((14336, 14336), 4096, (False, False))])
def test_group_of_one_row_products_equals_the_single_calls_bit_for_bit(L, t, ms, k, biased):
    rng = np.random.default_rng(len(ms) * 1000 + k + int(t))
    x = rng.standard_normal((1, k)).astype(np.float32)
    xd = _dev(x)
    ws_n = max(L.ggml_cdna4_mul_mat_workspace_size(int(t), k, 1), 256)
    ws = torch.empty(ws_n, dtype=torch.uint8, device="cuda")
    rb = R.row_size(t, k)
    wh = [R.random_weights(t, m, k, seed=11 * i + m) for i, m in enumerate(ms)]
    wd = [_dev(w) for w in wh]
    bias = [_dev(rng.standard_normal(m).astype(np.float32)) if hb else None for m, hb in zip(ms, biased)]
    single = [torch.empty((1, m), dtype=torch.float32, device="cuda") for m in ms]
    for w, m, y, bb in zip(wd, ms, single, bias):
        if bb is None:
            _ok(L, L.ggml_cdna4_mul_mat(int(t), w.data_ptr(), rb, xd.data_ptr(), k, y.data_ptr(), m, m, k, 1, ws.data_ptr(), ws_n, 0, 0, 0, _st()))
        else:
            _ok(L, L.ggml_cdna4_mul_mat_fused(int(t), w.data_ptr(), rb, xd.data_ptr(), k, y.data_ptr(), m, m, k, 1, bb.data_ptr(), 0, None, 0, ws.data_ptr(), ws_n, _st()))
    grouped = [torch.full((1, m), float("nan"), dtype=torch.float32, device="cuda") for m in ms]
    n = len(ms)
    rc = L.ggml_cdna4_mul_mat_group(int(t), n, _arr(C.c_void_p, [w.data_ptr() for w in wd]), _arr(C.c_int64, [rb] * n), _arr(C.c_int64, list(ms)),
                                    _arr(C.c_void_p, [y.data_ptr() for y in grouped]), _arr(C.c_void_p, [b.data_ptr() if b is not None else None for b in bias]),
                                    xd.data_ptr(), k, _st())
    _ok(L, rc)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, grouped)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (i, (a - b).abs().max().item())
    want = R.o_mul_mat(t, wh[0], x, ms[0], k)
    got = grouped[0].cpu().numpy() - (bias[0].cpu().numpy() if bias[0] is not None else 0.0)
    assert R.rel_l2(got, want) < 1e-5


def test_group_refuses_what_has_no_grouped_form(L):
    """more than four matrices, and an activation row that is not 16-byte aligned: -2, nothing launched, the outputs untouched"""
    t, k, m = R.Q4_K, 512, 64
    xd = _dev(np.ones((1, k + 8), np.float32))
    w = _dev(R.random_weights(t, m, k, seed=3))
    ys = [torch.full((1, m), 7.0, dtype=torch.float32, device="cuda") for _ in range(5)]
    rb = R.row_size(t, k)
    for n, xoff in ((5, 0), (2, 4)):
        rc = L.ggml_cdna4_mul_mat_group(int(t), n, _arr(C.c_void_p, [w.data_ptr()] * n), _arr(C.c_int64, [rb] * n), _arr(C.c_int64, [m] * n),
                                        _arr(C.c_void_p, [y.data_ptr() for y in ys[:n]]), _arr(C.c_void_p, [None] * n), xd.data_ptr() + xoff, k, _st())
        assert rc == -2, rc
    torch.cuda.synchronize()
    assert all(bool((y == 7.0).all()) for y in ys)
