"""-m gpu: the ONE-LAUNCH prefill step (round 5; k_gemm_kq_t64<.., FQ>, ggml_amd/csrc/gemm_kq_t64.inc) — activation quantizer -> grid barrier -> multiply inside one
kernel, what ggml_compute_forward_mul_mat does inside one op (src/ggml-cpu/ggml-cpu.c:7490-7509 then :7428-7605).

The bar is BIT-IDENTITY with the two launches it replaces (ggml_cdna4_prepare_act + ggml_cdna4_mul_mat_prepared: same quantized image, same sums in the same order),
plus the usual parity with the oracle, stability over repeated calls, HIP-graph replay and the routing contract (ggml_cdna4_mul_mat_route says 11 exactly where the
launch carries the quantizer; a shared device never takes it)."""
import numpy as np
import pytest
import torch
import refutil as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from ggml_amd import native, ops
    return native.lib(), native, ops


def _al256(n):
    return (n + 255) & ~255


def _image_view(ws, b, k):
    """the fp16 image inside a workspace of ggml_cdna4_mul_mat (capi.hip: carve: [qs int8][d f32][bsums i16][xh f16])"""
    off = _al256(b * k) + _al256(b * (k // 256) * 4) + _al256(b * (k // 16) * 2)
    return ws[off:off + b * k * 2]


def _one_launch(L, native, ops, a, x, ws, y, stream):
    m, k, b = a.M, a.K, x.shape[0]
    native.check(L.ggml_cdna4_mul_mat(int(a.type), a.data.data_ptr(), a.row_bytes, x.data_ptr(), x.stride(0), y.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, 0, 0, stream))


def _two_launches(L, native, ops, a, x, ws, y, stream, splitk=0):
    m, k, b = a.M, a.K, x.shape[0]
    native.check(L.ggml_cdna4_prepare_act(int(a.type), x.data_ptr(), x.stride(0), k, b, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, stream))
    native.check(L.ggml_cdna4_mul_mat_prepared(int(a.type), a.data.data_ptr(), a.row_bytes, y.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, 0, splitk, stream))


# (M, K, B) that take the one-launch step on a 256-CU part (a resident grid, the quantizer's share one pass at most and at least half of one — gemm_q_t64.hip:
# t64_fuses_quantizer): headline (split in two, hand-off inside the resident grid) | an odd superblock count 15 = 8 + 7 | deep split x 4 (two shapes) | unsplit, 256 tiles |
# ragged activation rows and weight rows
SHAPES = [(4096, 4096, 512), (4096, 3840, 512), (2048, 4096, 512), (4096, 8192, 256), (8192, 4096, 512), (4000, 4096, 500)]


@pytest.mark.parametrize("m,k,b", SHAPES)
def test_one_launch_step_is_bit_identical_to_quantize_then_gemm(env, m, k, b):
    L, native, ops = env
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the shapes are chosen for a 256-CU part")
    assert L.ggml_cdna4_mul_mat_route(int(R.Q4_K), m, k, b) == 11, "this shape should take the one-launch step"
    w = R.random_weights(R.Q4_K, m, k, seed=m + k)
    a = ops.QTensor.from_host_bytes(R.Q4_K, k, m, w, device="cuda:0")
    x = torch.from_numpy(np.random.default_rng(b).uniform(-1, 1, (b, k)).astype(np.float32)).cuda()
    x[min(3, b - 1), 256:512] = 0                                    # an all-zero superblock (d = 0)
    x[0, 17] = -9.5; x[0, 300] = 9.5                                 # equal |max|, opposite signs, different lanes of the 16-lane group: the first index wins
    st = torch.cuda.current_stream().cuda_stream
    n = L.ggml_cdna4_mul_mat_workspace_size(int(R.Q4_K), k, b)
    ws1 = torch.full((n,), 0x5A, dtype=torch.uint8, device="cuda"); ws2 = torch.full((n,), 0xA5, dtype=torch.uint8, device="cuda")
    y1 = torch.full((b, m), float("nan"), device="cuda"); y2 = torch.full((b, m), float("nan"), device="cuda")
    _one_launch(L, native, ops, a, x, ws1, y1, st)
    # the reference: two launches; the split of the launch above (AUTO inside a resident grid: the hand-off) is the explicit splitk = 2 of the prepared call where AUTO splits in two
    tiles = ((m + 127) // 128) * ((b + 127) // 128)
    _two_launches(L, native, ops, a, x, ws2, y2, st, splitk=2 if (tiles * 2 <= 256 and tiles * 4 > 256) else 0)
    torch.cuda.synchronize()
    img1, img2 = _image_view(ws1, b, k).cpu().numpy(), _image_view(ws2, b, k).cpu().numpy()
    assert np.array_equal(img1, img2), "the quantized fp16 image differs from k_quantize_q8_K's: %d bytes" % int((img1 != img2).sum())
    g1, g2 = y1.cpu().numpy(), y2.cpu().numpy()
    assert np.isfinite(g1).all()
    assert np.array_equal(g1.view(np.uint32), g2.view(np.uint32)), "one launch != two launches: max |diff| %.3e" % float(np.abs(g1 - g2).max())
    # and against the oracle on a row sample (the full-matrix compare of the headline shape lives in test_gpu_parity.py)
    rows = np.unique(np.concatenate([np.arange(0, min(64, m)), np.random.default_rng(1).integers(0, m, 64)]))
    want = R.o_mul_mat(R.Q4_K, np.ascontiguousarray(w.reshape(m, -1)[rows]).reshape(-1), x.cpu().numpy(), len(rows), k)
    assert R.rel_l2(g1[:, rows], want) < 1e-3


def test_one_launch_step_is_stable_over_200_calls_and_replays_from_a_graph(env):
    L, native, ops = env
    m, k, b = 4096, 4096, 512
    if L.ggml_cdna4_mul_mat_route(int(R.Q4_K), m, k, b) != 11:
        pytest.skip("not a part on which the headline shape takes the one-launch step")
    w = R.random_weights(R.Q4_K, m, k, seed=7)
    a = ops.QTensor.from_host_bytes(R.Q4_K, k, m, w, device="cuda:0")
    rng = np.random.default_rng(11)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (b, k)).astype(np.float32)).cuda() for _ in range(2)]
    n = L.ggml_cdna4_mul_mat_workspace_size(int(R.Q4_K), k, b)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    first = []
    for x in xs:
        y = torch.empty((b, m), device="cuda")
        _one_launch(L, native, ops, a, x, ws, y, st)
        first.append(y.clone())
    # 200 calls alternating between two activation sets on ONE workspace: a stale image line (a missed write-through / a load served by a stale cache line) or a barrier
    # that lets a work-group through early shows up as a difference
    y = torch.empty((b, m), device="cuda")
    bad = 0
    for i in range(200):
        _one_launch(L, native, ops, a, xs[i & 1], ws, y, st)
        bad += int((y.view(torch.int32) != first[i & 1].view(torch.int32)).sum().item())
    assert bad == 0, "%d differing outputs over 200 calls" % bad
    # HIP-graph replay: the barrier's words and the exchange flags are back at rest after every launch, no per-launch state on the host
    side = torch.cuda.Stream()
    yg = torch.empty((b, m), device="cuda")
    with torch.cuda.stream(side):
        _one_launch(L, native, ops, a, xs[0], ws, yg, side.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(4):
            _one_launch(L, native, ops, a, xs[0], ws, yg, torch.cuda.current_stream().cuda_stream)
    for _ in range(25):
        yg.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(yg.view(torch.int32), first[0].view(torch.int32))


def test_a_shared_device_never_takes_the_one_launch_step(env):
    L, native, ops = env
    m, k, b = 4096, 4096, 512
    if L.ggml_cdna4_mul_mat_route(int(R.Q4_K), m, k, b) != 11:
        pytest.skip("not a part on which the headline shape takes the one-launch step")
    old = L.ggml_cdna4_set_shared_device(1)
    try:
        assert L.ggml_cdna4_mul_mat_route(int(R.Q4_K), m, k, b) == 10            # quantize + k_gemm_kq_t64 with the ticketed split: nobody waits for anybody
        w = R.random_weights(R.Q4_K, m, k, seed=3)
        a = ops.QTensor.from_host_bytes(R.Q4_K, k, m, w, device="cuda:0")
        x = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (b, k)).astype(np.float32)).cuda()
        y_shared = ops.mul_mat(a, x, path=ops.PATH_GEMM)
    finally:
        L.ggml_cdna4_set_shared_device(old)
    y_owned = ops.mul_mat(a, x, path=ops.PATH_GEMM)
    torch.cuda.synchronize()
    # two fp32 partial sums per element either way (commutative): the modes agree bit for bit at an even superblock count
    assert torch.equal(y_shared.view(torch.int32), y_owned.view(torch.int32))


@pytest.mark.parametrize("tail", ["bias", "bias_gelu", "bias_resid"])
def test_one_launch_step_carries_the_tail(env, tail):
    """ggml_cdna4_mul_mat_fused at the headline shape: ONE launch for quantizer + product + bias / GELU / residual, bit-identical to the same tail behind the two-launch product"""
    L, native, ops = env
    m, k, b = 4096, 4096, 512
    if L.ggml_cdna4_mul_mat_route(int(R.Q4_K), m, k, b) != 11:
        pytest.skip("not a part on which the headline shape takes the one-launch step")
    w = R.random_weights(R.Q4_K, m, k, seed=9)
    a = ops.QTensor.from_host_bytes(R.Q4_K, k, m, w, device="cuda:0")
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.uniform(-1, 1, (b, k)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.uniform(-1, 1, (m,)).astype(np.float32)).cuda()
    resid = torch.from_numpy(rng.uniform(-1, 1, (b, m)).astype(np.float32)).cuda()
    n = L.ggml_cdna4_mul_mat_workspace_size(int(R.Q4_K), k, b)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    act = 1 if tail == "bias_gelu" else 0
    rp = resid.data_ptr() if tail == "bias_resid" else None
    yf = torch.empty((b, m), device="cuda")
    native.check(L.ggml_cdna4_mul_mat_fused(int(R.Q4_K), a.data.data_ptr(), a.row_bytes, x.data_ptr(), k, yf.data_ptr(), m, m, k, b, bias.data_ptr(), act, rp, m, ws.data_ptr(), ws.numel(), st))
    old = L.ggml_cdna4_set_shared_device(1)                              # the two-launch route (ticketed split), same tail in the store
    try:
        yr = torch.empty((b, m), device="cuda")
        native.check(L.ggml_cdna4_mul_mat_fused(int(R.Q4_K), a.data.data_ptr(), a.row_bytes, x.data_ptr(), k, yr.data_ptr(), m, m, k, b, bias.data_ptr(), act, rp, m, ws.data_ptr(), ws.numel(), st))
    finally:
        L.ggml_cdna4_set_shared_device(old)
    torch.cuda.synchronize()
    assert torch.isfinite(yf).all()
    assert torch.equal(yf.view(torch.int32), yr.view(torch.int32))
