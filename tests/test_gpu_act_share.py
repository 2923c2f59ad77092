"""-m gpu: the hand-off of quantized activations between MUL_MATs that read the same src1 (round 5).  A transformer layer multiplies one activation matrix
by wq / wk / wv and another by w_gate / w_up; the CPU backend quantizes src1 once per MUL_MAT node (src/ggml-cpu/ggml-cpu.c:7490-7509) and so did every ggml_cdna4_mul_mat call.
C-ABI: ggml_cdna4_act_image_key names the image a call leaves in its workspace; ggml_cdna4_mul_mat_prepared[_fused] multiply it.  Plug-in: the graph walk takes the hand-off
whenever src1 and the workspace are untouched since the previous MUL_MAT (oracle/split_harness.cpp `shared`, through ggml's public API).  The bar is BIT-IDENTITY with the
calls that quantize again."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest
import torch
import refutil as R

pytestmark = pytest.mark.gpu
PLUGIN = os.path.join(R.ROOT, "ggml_amd", "lib", "libggml-cdna4.so")
EXE = os.path.join(R.REF_DIR, "split_harness")


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from ggml_amd import native, ops
    return native.lib(), native, ops


@pytest.mark.parametrize("b", [16, 96, 512])
@pytest.mark.parametrize("t1,t2", [(R.Q4_K, R.Q6_K), (R.Q4_0, R.Q8_0), (R.Q5_K, R.Q4_K), (R.Q2_K, R.IQ4_XS)])
def test_second_product_on_the_first_ones_image_is_bit_identical(env, t1, t2, b):
    """W1 . X through ggml_cdna4_mul_mat, then W2 . X (another format of the same activation class, another M) through ggml_cdna4_mul_mat_prepared / _prepared_fused on the
    workspace the first call left — where ggml_cdna4_act_image_key says the two calls build the same image; equal to ggml_cdna4_mul_mat / _mul_mat_fused of W2, bit for bit"""
    from test_gpu_cabi_ops import _dev, _ok, _st
    L, native, ops = env
    k, m1, m2 = 1024, 1024, 384
    k1, k2 = L.ggml_cdna4_act_image_key(int(t1), m1, k, b), L.ggml_cdna4_act_image_key(int(t2), m2, k, b)
    if k1 == 0 or k1 != k2:
        pytest.skip("these two calls do not build the same image (keys %d / %d): nothing to hand off" % (k1, k2))
    w1, w2 = R.random_weights(t1, m1, k, seed=1), R.random_weights(t2, m2, k, seed=2)
    rng = np.random.default_rng(b)
    x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
    bias = rng.standard_normal(m2).astype(np.float32)
    res = rng.standard_normal((b, m2)).astype(np.float32)
    w1d, w2d, xd, bd, rd = _dev(w1), _dev(w2), _dev(x), _dev(bias), _dev(res)
    nws = max(L.ggml_cdna4_mul_mat_workspace_size(int(t1), k, b), L.ggml_cdna4_mul_mat_workspace_size(int(t2), k, b), 256)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    y1 = torch.empty((b, m1), dtype=torch.float32, device="cuda")
    ya = torch.empty((b, m2), dtype=torch.float32, device="cuda"); yb = torch.full((b, m2), 3.0, dtype=torch.float32, device="cuda"); yc = torch.empty_like(ya)
    yd = torch.full((b, m2), 5.0, dtype=torch.float32, device="cuda")
    rb1, rb2 = R.row_size(t1, k), R.row_size(t2, k)
    # the calls that quantize again
    _ok(L, L.ggml_cdna4_mul_mat(int(t2), w2d.data_ptr(), rb2, xd.data_ptr(), k, ya.data_ptr(), m2, m2, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, _st()))
    _ok(L, L.ggml_cdna4_mul_mat_fused(int(t2), w2d.data_ptr(), rb2, xd.data_ptr(), k, yc.data_ptr(), m2, m2, k, b, bd.data_ptr(), 1, None, 0, ws.data_ptr(), ws.numel(), _st()))
    ws.zero_()
    # W1 . X leaves the image; W2 multiplies it
    _ok(L, L.ggml_cdna4_mul_mat(int(t1), w1d.data_ptr(), rb1, xd.data_ptr(), k, y1.data_ptr(), m1, m1, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, _st()))
    _ok(L, L.ggml_cdna4_mul_mat_prepared(int(t2), w2d.data_ptr(), rb2, yb.data_ptr(), m2, m2, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, _st()))
    _ok(L, L.ggml_cdna4_mul_mat_prepared_fused(int(t2), w2d.data_ptr(), rb2, yd.data_ptr(), m2, m2, k, b, bd.data_ptr(), 1, None, 0, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    assert torch.equal(ya.view(torch.int32), yb.view(torch.int32))
    assert torch.equal(yc.view(torch.int32), yd.view(torch.int32))
    assert R.rel_l2(yb.cpu().numpy(), R.o_mul_mat(t2, w2, x, m2, k)) < 1e-3
    assert R.rel_l2(y1.cpu().numpy(), R.o_mul_mat(t1, w1, x, m1, k)) < 1e-3
    # + residual through the prepared twin
    ye = torch.empty_like(ya); yf = torch.full((b, m2), 9.0, dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_mul_mat_fused(int(t2), w2d.data_ptr(), rb2, xd.data_ptr(), k, ye.data_ptr(), m2, m2, k, b, bd.data_ptr(), 0, rd.data_ptr(), m2, ws.data_ptr(), ws.numel(), _st()))
    _ok(L, L.ggml_cdna4_mul_mat_prepared_fused(int(t2), w2d.data_ptr(), rb2, yf.data_ptr(), m2, m2, k, b, bd.data_ptr(), 0, rd.data_ptr(), m2, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    assert torch.equal(ye.view(torch.int32), yf.view(torch.int32))


@pytest.mark.parametrize("rms,affine", [(1, "gain"), (0, "gain_shift"), (1, "none")])
@pytest.mark.parametrize("t,m,k,b", [(R.Q4_K, 768, 1024, 96), (R.Q6_K, 256, 512, 200), (R.Q5_K, 1024, 2048, 130), (R.Q4_K, 512, 8192, 72),
                                     (R.Q4_0, 2304, 768, 200), (R.Q8_0, 256, 1280, 96), (R.Q5_0, 512, 1024, 130)])
def test_norm_that_also_leaves_the_activation_image(env, t, m, k, b, rms, affine):
    """ggml_cdna4_op_norm_affine_q8_K: the normalised rows (same bits as ggml_cdna4_op_norm_affine) AND, in the workspace, the image ggml_cdna4_mul_mat would build of them —
    ggml_cdna4_mul_mat_prepared on that workspace equals ggml_cdna4_mul_mat on the normalised rows, bit for bit (the graph's first reader of a normalised tensor pays no
    quantizer launch: plug-in try_fused_norm)"""
    from test_gpu_cabi_ops import _desc, _dev, _ok, _st
    L, native, ops = env
    kq = t in (R.Q4_K, R.Q5_K, R.Q6_K)
    assert L.ggml_cdna4_act_image_key(int(t), m, k, b) == (19 if kq else 17)
    norm_act = L.ggml_cdna4_op_norm_affine_q8_K if kq else L.ggml_cdna4_op_norm_affine_q8_0              # (Q8_0-class formats, round 5: gpt-2's Q4_0 layers at prompt sizes)
    rng = np.random.default_rng(k + b)
    x = (rng.standard_normal((b, k)) * 3 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(k)).astype(np.float32); sh = (0.05 * rng.standard_normal(k)).astype(np.float32)
    w = R.random_weights(t, m, k, seed=3)
    xd, gd, sd, wd = _dev(x), _dev(g), _dev(sh), _dev(w)
    y_ref = torch.empty((b, k), dtype=torch.float32, device="cuda"); y_q = torch.empty((b, k), dtype=torch.float32, device="cuda")
    gp = C.byref(_desc(gd, R.F32)) if affine != "none" else None
    sp = C.byref(_desc(sd, R.F32)) if affine == "gain_shift" else None
    ws = torch.empty(max(L.ggml_cdna4_mul_mat_workspace_size(int(t), k, b), 256), dtype=torch.uint8, device="cuda")
    _ok(L, L.ggml_cdna4_op_norm_affine(C.byref(_desc(xd, R.F32)), gp, sp, C.byref(_desc(y_ref, R.F32)), 1e-5, rms, _st()))
    _ok(L, norm_act(C.byref(_desc(xd, R.F32)), gp, sp, C.byref(_desc(y_q, R.F32)), 1e-5, rms, int(t), ws.data_ptr(), ws.numel(), _st()))
    rb = R.row_size(t, k)
    o_prep = torch.full((b, m), 2.0, dtype=torch.float32, device="cuda"); o_ref = torch.empty((b, m), dtype=torch.float32, device="cuda")
    _ok(L, L.ggml_cdna4_mul_mat_prepared(int(t), wd.data_ptr(), rb, o_prep.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, _st()))
    ws2 = torch.empty_like(ws)
    _ok(L, L.ggml_cdna4_mul_mat(int(t), wd.data_ptr(), rb, y_ref.data_ptr(), k, o_ref.data_ptr(), m, m, k, b, ws2.data_ptr(), ws2.numel(), 0, 0, 0, _st()))
    torch.cuda.synchronize()
    assert torch.equal(y_q.view(torch.int32), y_ref.view(torch.int32))
    assert torch.equal(o_prep.view(torch.int32), o_ref.view(torch.int32))
    assert R.rel_l2(o_ref.cpu().numpy(), R.o_mul_mat(t, w, y_ref.cpu().numpy(), m, k)) < 1e-3
    # refusals: the other activation class
    other = L.ggml_cdna4_op_norm_affine_q8_0 if kq else L.ggml_cdna4_op_norm_affine_q8_K
    assert other(C.byref(_desc(xd, R.F32)), gp, sp, C.byref(_desc(y_q, R.F32)), 1e-5, rms, int(t), ws.data_ptr(), ws.numel(), _st()) != 0


@pytest.mark.parametrize("t,k,b", [(R.Q4_0, 800, 5), (R.Q4_0, 96, 3), (R.Q8_0, 1056, 4), (R.Q4_0, 768, 200), (R.Q4_K, 768, 9), (R.Q6_K, 2048, 33), (R.Q5_K, 8192, 3)])
def test_the_image_a_norm_leaves_is_prepare_acts_image_byte_for_byte(env, t, k, b):
    """ggml_cdna4_op_norm_affine_q8_K / _q8_0 against ggml_cdna4_op_norm_affine + ggml_cdna4_prepare_act(PATH_GEMM): the fp32 rows and every byte of the fp16 image in the
    workspace (capi.hip: carve — [qs][d][bsums][xh]); rows whose groups of four do not fill the last wave (800 = 200 groups, 96 = 24), a ragged last 128-panel (1056)"""
    from test_gpu_cabi_ops import _desc, _dev, _ok, _st
    L, native, ops = env
    kq = t in (R.Q4_K, R.Q5_K, R.Q6_K)
    rng = np.random.default_rng(k + b)
    x = _dev((rng.standard_normal((b, k)) * 2).astype(np.float32)); g = _dev((1 + 0.1 * rng.standard_normal(k)).astype(np.float32))
    y1 = torch.empty((b, k), dtype=torch.float32, device="cuda"); y2 = torch.empty((b, k), dtype=torch.float32, device="cuda")
    n = max(L.ggml_cdna4_mul_mat_workspace_size(int(t), k, b), 256)
    ws1 = torch.zeros(n, dtype=torch.uint8, device="cuda"); ws2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    _ok(L, L.ggml_cdna4_op_norm_affine(C.byref(_desc(x, R.F32)), C.byref(_desc(g, R.F32)), None, C.byref(_desc(y1, R.F32)), 1e-5, 1, _st()))
    _ok(L, L.ggml_cdna4_prepare_act(int(t), y1.data_ptr(), k, k, b, ws1.data_ptr(), ws1.numel(), 2, _st()))
    fn = L.ggml_cdna4_op_norm_affine_q8_K if kq else L.ggml_cdna4_op_norm_affine_q8_0
    _ok(L, fn(C.byref(_desc(x, R.F32)), C.byref(_desc(g, R.F32)), None, C.byref(_desc(y2, R.F32)), 1e-5, 1, int(t), ws2.data_ptr(), ws2.numel(), _st()))
    torch.cuda.synchronize()
    a256 = lambda v: (v + 255) // 256 * 256
    off = a256(b * k) + a256(b * (k // (256 if kq else 32)) * 4) + a256(b * (k // 16) * 2)
    nimg = b * ((k + 127) // 128 * 128) * 2
    assert torch.equal(y1.view(torch.int32), y2.view(torch.int32))
    img1, img2 = ws1[off:off + nimg].cpu().numpy(), ws2[off:off + nimg].cpu().numpy()
    assert (img1 != 0).sum() > nimg // 4 and np.array_equal(img1, img2)


def test_prepared_fused_refuses_a_shape_without_an_image(env):
    from test_gpu_cabi_ops import _dev, _st
    L, native, ops = env
    m, k = 256, 512
    assert L.ggml_cdna4_act_image_key(int(R.Q4_K), m, k, 1) == 0
    w, x, bias = _dev(R.random_weights(R.Q4_K, m, k, seed=1)), _dev(np.zeros((1, k), np.float32)), _dev(np.zeros(m, np.float32))
    y = torch.empty((1, m), dtype=torch.float32, device="cuda"); ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    assert L.ggml_cdna4_mul_mat_prepared_fused(int(R.Q4_K), w.data_ptr(), R.row_size(R.Q4_K, k), y.data_ptr(), m, m, k, 1, bias.data_ptr(), 0, None, 0, ws.data_ptr(), ws.numel(), _st()) != 0
    assert b"no prepared form" in L.ggml_cdna4_last_error()


def _harness(type_, d, h, b, share, exact=False):
    if not os.path.exists(EXE):
        pytest.fail("prebuilt oracle/_ref/split_harness missing from the snapshot")
    e = dict(os.environ)
    e.pop("GGML_CDNA4_NO_ACT_SHARE", None)
    e.pop("GGML_CDNA4_EXACT", None)
    if not share:
        e["GGML_CDNA4_NO_ACT_SHARE"] = "1"
    if exact:
        e["GGML_CDNA4_EXACT"] = "1"
        e["HARNESS_NO_TIMING"] = "1"
    r = subprocess.run([EXE, PLUGIN, type_, str(d), str(h), str(b), "shared"], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    j["share"] = share
    j["exact"] = exact
    os.makedirs(os.path.join(R.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(R.ROOT, "gpurun_out", "split_report.jsonl"), "a") as f:
        f.write(json.dumps(j) + "\n")
    return j


@pytest.mark.parametrize("type_,d,h,b,hand_offs", [("q4_K", 1024, 2816, 96, 5), ("q4_K", 1024, 2816, 16, 3), ("q4_0", 768, 3072, 128, 5), ("q6_K", 1024, 2048, 200, 5),
                                                    ("q8_0", 512, 2048, 24, 3), ("q4_K", 1024, 2816, 1, 0), ("q4_K", 2048, 5632, 512, 5), ("q5_K", 4096, 4096, 64, 3)])       # (Q5_K at 64 rows: the int8 matrix cores — their image is not the one a NORM chain leaves)
def test_a_layers_shared_activations_are_quantized_once_through_ggmls_public_api(type_, d, h, b, hand_offs):
    """rms_norm -> {wq, wk, wv + bias} and rms_norm -> {w_gate, w_up} -> w_down on the plug-in: three of the six MUL_MATs multiply the previous one's image — FIVE on the fp16 GEMM routes
    (K-quants: the Q8_K image; Q4_0 / Q8_0: the Q8_0 image), where the rms_norm chain's own launch leaves the image and wq / w_gate take the hand-off too (none at decode size, where the quantizer lives inside
    the GEMV launch).  THE GATE: the outputs' bytes equal those of a run with the hand-off off.  Against the CPU backend: K and V — one product of the norm's output — within north_star's
    per-product 1e-3.  `out` sits behind two RE-QUANTIZATIONS of computed activations, where a last-bit difference in Q moves int8 steps of the next product's input (the CPU algorithm's own
    sensitivity to its inputs): its distance is REPORTED (gpurun_out/split_report.jsonl), not asserted — what is asserted for the chain is the reference-order run below, which is bit-identical."""
    on, off = _harness(type_, d, h, b, True), _harness(type_, d, h, b, False)
    assert on["act_hand_offs_first_compute"] == hand_offs and off["act_hand_offs_first_compute"] == 0, (on, off)
    assert on["fnv1a"] == off["fnv1a"], (on, off)
    assert on["k_vs_cpu"] < 1e-3 and on["v_vs_cpu"] < 1e-3, on


@pytest.mark.parametrize("type_,d,h,b", [("q4_K", 1024, 2816, 96), ("q4_K", 1024, 2816, 1), ("q5_K", 1024, 2048, 24), ("q6_K", 1024, 2048, 40), ("q4_0", 768, 3072, 33), ("q8_0", 512, 2048, 24)])
def test_the_layer_front_in_reference_order_is_the_cpu_backends_bits(type_, d, h, b):
    """GGML_CDNA4_EXACT=1 on the same attention + FFN front: RMS_NORM, the six quantized MUL_MATs (K-quants: ggml_vec_dot_q4_K_q8_K / _q5_K_q8_K / _q6_K_q8_K in their AVX2 lane
    order), SILU, MUL and ADD — every fp32 word of K, V and out (three products and two re-quantizations deep) has the CPU backend's bits."""
    j = _harness(type_, d, h, b, True, exact=True)
    assert j["act_hand_offs_first_compute"] == 0 and j["grouped_first_compute"] == 0, j          # (the mode runs node by node)
    assert j["words_differing_from_cpu"] == 0 and j["out_vs_cpu"] == 0.0, j


def _moeffn(type_, d, h, tokens, share):
    if not os.path.exists(EXE):
        pytest.fail("prebuilt oracle/_ref/split_harness missing from the snapshot")
    e = dict(os.environ)
    e.pop("GGML_CDNA4_NO_ACT_SHARE", None)
    if not share:
        e["GGML_CDNA4_NO_ACT_SHARE"] = "1"
    r = subprocess.run([EXE, PLUGIN, type_, str(d), str(h), str(tokens), "moeffn"], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    j["share"] = share
    os.makedirs(os.path.join(R.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(R.ROOT, "gpurun_out", "split_report.jsonl"), "a") as f:
        f.write(json.dumps(j) + "\n")
    return j


@pytest.mark.parametrize("type_,d,h,tokens", [("q4_K", 2048, 1536, 512), ("q4_K", 1024, 2048, 100)])
def test_moe_ffn_gate_stack_multiplies_the_up_stacks_front(type_, d, h, tokens):
    """a mixture-of-experts FFN through ggml's public API (8 experts, 2 used; up, gate = MUL_MAT_ID of the same (cur, ids); down behind silu(gate) * up): the second stack takes the
    first one's front — one launch instead of two —, THE GATE being byte equality with the sharing off; the distance to the CPU backend (two grouped fp16 products and a
    re-quantization deep) is reported"""
    on, off = _moeffn(type_, d, h, tokens, True), _moeffn(type_, d, h, tokens, False)
    assert on["moe_fronts_shared_first_compute"] == 1 and off["moe_fronts_shared_first_compute"] == 0, (on, off)
    assert on["fnv1a"] == off["fnv1a"], (on, off)
