"""-m gpu: MUL_MAT_IDs of one (b, ids) share the first one's FRONT — ggml_cdna4_mul_mat_id_front_key / ggml_cdna4_mul_mat_id_prepared through the C-ABI: the up- and
gate-projection expert stacks of a mixture-of-experts layer (llama.cpp build_moe_ffn; the reference sorts and quantizes once per node, ggml-cpu.c:7609-7784).  The prepared call —
ONE launch — is bit-identical to the full call, and calls that take another route have no key and are refused with -2."""
import numpy as np
import pytest
import torch

import refutil as R
from test_gpu_cabi_ops import L, _dev, _ok, _st      # noqa: F401  (the fixture + helpers)

pytestmark = pytest.mark.gpu


def _args(t, w, rb, m, x, k, ids, y, n_expert, n_used, n_b, n_tok, ws):
    return (int(t), w.data_ptr(), rb, m * rb, x.data_ptr(), k, n_b * k, ids.data_ptr(), n_used, y.data_ptr(), m, n_used * m, m, k, n_expert, n_used, n_b, n_tok, ws.data_ptr(), ws.numel(), _st())


@pytest.mark.parametrize("n_expert,n_used,n_b_is_one,n_tok,m,k", [(4, 2, True, 96, 256, 512), (8, 2, True, 512, 1024, 1024), (8, 2, False, 40, 384, 256), (3, 1, True, 300, 200, 768)])
def test_second_stack_multiplies_the_first_ones_front_bit_for_bit(L, n_expert, n_used, n_b_is_one, n_tok, m, k):
    t = R.Q4_K
    rng = np.random.default_rng(n_expert * 100 + n_tok + m)
    n_b = 1 if n_b_is_one else n_used
    rb = R.row_size(t, k)
    w1, w2 = _dev(R.random_weights(t, n_expert * m, k, seed=5)), _dev(R.random_weights(t, n_expert * m, k, seed=6))
    x = _dev(rng.standard_normal((n_tok, n_b, k)).astype(np.float32))
    ids = _dev(np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32))
    nws = L.ggml_cdna4_mul_mat_id_workspace_size(int(t), k, n_expert, n_used, n_b, n_tok)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    key = L.ggml_cdna4_mul_mat_id_front_key(int(t), w1.data_ptr(), rb, m * rb, m, k, n_expert, n_used, n_b, n_tok, nws)
    assert key != 0 and key == L.ggml_cdna4_mul_mat_id_front_key(int(t), w2.data_ptr(), rb, m * rb, m, k, n_expert, n_used, n_b, n_tok, nws)
    y1, y2, y2_full = (torch.full((n_tok, n_used, m), float("nan"), dtype=torch.float32, device="cuda") for _ in range(3))
    _ok(L, L.ggml_cdna4_mul_mat_id(*_args(t, w2, rb, m, x, k, ids, y2_full, n_expert, n_used, n_b, n_tok, ws)))
    ws.zero_()
    _ok(L, L.ggml_cdna4_mul_mat_id(*_args(t, w1, rb, m, x, k, ids, y1, n_expert, n_used, n_b, n_tok, ws)))
    _ok(L, L.ggml_cdna4_mul_mat_id_prepared(*_args(t, w2, rb, m, x, k, ids, y2, n_expert, n_used, n_b, n_tok, ws)))
    torch.cuda.synchronize()
    assert torch.equal(y2.view(torch.int32), y2_full.view(torch.int32)), (y2 - y2_full).abs().max().item()
    assert not torch.equal(y1.view(torch.int32), y2.view(torch.int32))


def test_calls_on_other_routes_leave_no_front(L):
    """one token (the one-launch decode), a format on the per-tile route (Q6_K), a workspace too small for the queue's tables: key 0, and the prepared call answers -2 without a launch"""
    k, m, ne, nu = 512, 128, 4, 2
    for t, n_tok, shrink in ((R.Q4_K, 1, 0), (R.Q6_K, 96, 0), (R.Q4_K, 96, 1)):
        rb = R.row_size(t, k)
        w = _dev(R.random_weights(t, ne * m, k, seed=9))
        nws = L.ggml_cdna4_mul_mat_id_workspace_size(int(t), k, ne, nu, 1, n_tok)
        use = 4096 if shrink else nws
        assert L.ggml_cdna4_mul_mat_id_front_key(int(t), w.data_ptr(), rb, m * rb, m, k, ne, nu, 1, n_tok, use) == 0
        x = _dev(np.ones((n_tok, 1, k), np.float32)); ids = _dev(np.zeros((n_tok, nu), np.int32) + np.arange(nu, dtype=np.int32))
        y = torch.full((n_tok, nu, m), 3.0, dtype=torch.float32, device="cuda")
        ws = torch.empty(max(nws, 4096), dtype=torch.uint8, device="cuda")
        a = list(_args(t, w, rb, m, x, k, ids, y, ne, nu, 1, n_tok, ws)); a[-2] = use
        assert L.ggml_cdna4_mul_mat_id_prepared(*a) == -2
        torch.cuda.synchronize()
        assert bool((y == 3.0).all())
