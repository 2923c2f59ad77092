"""-m gpu parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.
Bars: bit-exact for the activation quantizers; rel-L2 <= 1e-5 for the int8-dot GEMV path (exact integer block
sums, only fp32 summation order differs — measured ~2e-7); rel-L2 <= 1e-3 for the fp16-MFMA GEMM path (the
tolerance BASELINE.json's north_star states; measured ~3e-4).  Norm-wise, never element-wise."""
import os
import numpy as np
import pytest
import torch
import refutil as R

pytestmark = pytest.mark.gpu

WT = list(R.QUANT_TYPES.items())
TOL_GEMV, TOL_GEMM = 1e-5, 1e-3


@pytest.fixture(scope="module")
def gu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    import gpu_util
    from ggml_amd import native
    native.lib()
    return gpu_util


def _x(seed, b, k, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
    elif kind == "normal":
        x = (rng.standard_normal((b, k)) * 3).astype(np.float32)
    elif kind == "ties":
        x = (rng.integers(-254, 255, (b, k)) / 2.0).astype(np.float32)
    else:
        x = (0.1 + 2 * np.cos(np.arange(b * k, dtype=np.float32) + seed)).reshape(b, k).astype(np.float32)
    return x


def _same_f16(got, want):
    """fp16 images must agree value for value (+0 == -0: the sign of a zero product is not meaningful)"""
    g, w = got.astype(np.float32), want.astype(np.float32)
    bad = np.argwhere(g != w)
    assert bad.size == 0, "%d mismatches, first at %s: got %r want %r" % (len(bad), bad[0], g[tuple(bad[0])], w[tuple(bad[0])])


# ------------------------------------------------------------------------------------------------ quantizers
@pytest.mark.parametrize("kind", ["uniform", "normal", "ties", "cos"])
def test_quantize_q8_K_bit_exact(gu, kind):
    from ggml_amd import ops
    x = _x(5, 12, 2048, kind)
    x[3, 256:512] = 0
    x[4, 17] = -9.5; x[4, 300] = 9.5          # equal |max| with opposite signs: first index wins
    x[5, 700] = 7.25; x[5, 701] = -7.25
    qs, d, bs, xh = [t.cpu().numpy() for t in ops.quantize_row_q8_K(gu.to_dev(x), want_f16=True)]
    ref = R.o_quantize_act(R.Q4_K, x).reshape(12, -1, 292)
    rd = ref[:, :, 0:4].copy().view(np.float32).reshape(12, -1)
    rq = ref[:, :, 4:260].copy().view(np.int8).reshape(12, -1)
    rb = ref[:, :, 260:292].copy().view(np.int16).reshape(12, -1)
    assert np.array_equal(d.view(np.uint32), rd.view(np.uint32))
    assert np.array_equal(qs, rq)
    assert np.array_equal(bs, rb)
    want = (np.repeat(rd, 256, axis=1) * rq.astype(np.float32)).astype(np.float16)
    _same_f16(gu.uninterleave(xh), want)


@pytest.mark.parametrize("ref_rounding", [False, True])
@pytest.mark.parametrize("kind", ["uniform", "normal", "ties"])
def test_quantize_q8_0_bit_exact(gu, kind, ref_rounding):
    from ggml_amd import ops
    x = _x(6, 9, 1024, kind)
    x[2, 32:64] = 0
    qs, d, xh = [t.cpu().numpy() for t in ops.quantize_row_q8_0(gu.to_dev(x), ref_rounding=ref_rounding, want_f16=True)]
    name = "q8_0_ref" if ref_rounding else "q8_0_cpu"
    ref = np.stack([R.o_quantize_row(name, x[i]) for i in range(9)]).reshape(9, -1, 34)
    rd = ref[:, :, 0:2].copy().view(np.float16).astype(np.float32).reshape(9, -1)
    rq = ref[:, :, 2:34].copy().view(np.int8).reshape(9, -1)
    assert np.array_equal(qs, rq)
    assert np.array_equal(d, rd)
    want = (np.repeat(rd, 32, axis=1) * rq.astype(np.float32)).astype(np.float16)
    _same_f16(gu.uninterleave(xh), want)


@pytest.mark.parametrize("kind", ["uniform", "normal", "ties", "cos"])
def test_quantize_q8_1_bit_exact(gu, kind):
    """block_q8_1 of the CPU backend (AVX2 body, ggml-cpu-quants.c:1076-1119) bit for bit: quants, d, and s = fp16(fp32(d * sum q)) — the
    product rounded to fp32 FIRST (round 2: hipcc folded the multiply into the conversion, one rounding, one fp16 ulp off in ~1 block of 256).
    8192 blocks per case so that a folded conversion cannot hide."""
    from ggml_amd import ops
    x = _x(7, 64, 4096, kind)
    x[2, 32:64] = 0
    qs, d, s, xh = [t.cpu().numpy() for t in ops.quantize_row_q8_1(gu.to_dev(x), want_f16=True)]
    ref = R.o_quantize_act(R.Q4_1, x).reshape(64, -1, 36)
    rd = ref[:, :, 0:2].copy().view(np.float16).astype(np.float32).reshape(64, -1)
    rs = ref[:, :, 2:4].copy().view(np.float16).astype(np.float32).reshape(64, -1)
    rq = ref[:, :, 4:36].copy().view(np.int8).reshape(64, -1)
    assert np.array_equal(qs, rq)
    assert np.array_equal(d, rd)
    bad = np.argwhere(s != rs)
    assert bad.size == 0, "%d of %d s values differ, first at %s: got %r want %r" % (len(bad), s.size, bad[0], s[tuple(bad[0])], rs[tuple(bad[0])])
    want = (np.repeat(rd, 32, axis=1) * rq.astype(np.float32)).astype(np.float16)
    _same_f16(gu.uninterleave(xh), want)


# ------------------------------------------------------------------------------------------------ golden
@pytest.mark.parametrize("name,t", WT)
def test_golden_fixture_gemv_and_gemm(gu, name, t):
    """committed fixtures produced by the unmodified reference (tests/golden/make_golden.py)"""
    from ggml_amd import ops
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mul_mat_small.npz"))
    M, K = int(G["M"]), int(G["K"])
    a = gu.qtensor(t, G[name + "_w"], M, K)
    x = G[name + "_x"]
    y = ops.mul_mat(a, gu.to_dev(x), path=ops.PATH_GEMV).cpu().numpy()
    e = R.rel_l2(y, G[name + "_y"]); gu.report(test="golden_gemv", type=name, rel_l2=e)
    assert e < TOL_GEMV
    x16 = np.concatenate([x] * 4)
    y = ops.mul_mat(a, gu.to_dev(x16), path=ops.PATH_GEMM).cpu().numpy()
    e = R.rel_l2(y, np.concatenate([G[name + "_y"]] * 4)); gu.report(test="golden_gemm", type=name, rel_l2=e)
    assert e < TOL_GEMM


def test_golden_mul_mat_id(gu):
    from ggml_amd import ops
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mul_mat_id_small.npz"))
    for name in R.QUANT_TYPES:
        t = R.QUANT_TYPES[name]
        M, K, ne = int(G["M"]), int(G["K"]), int(G["n_expert"])
        a = gu.qtensor(t, G[name + "_w"], ne * M, K)
        y = ops.mul_mat_id(a, gu.to_dev(G[name + "_x"]), gu.to_dev(G["ids"]), n_expert=ne).cpu().numpy()
        e = R.rel_l2(y, G[name + "_y"]); gu.report(test="golden_mul_mat_id", type=name, rel_l2=e)
        assert e < TOL_GEMV
        # the single-token (one-launch) path against the same fixture
        y1 = ops.mul_mat_id(a, gu.to_dev(G[name + "_x"][:1]), gu.to_dev(G["ids"][:1]), n_expert=ne).cpu().numpy()
        assert R.rel_l2(y1, G[name + "_y"][:1]) < TOL_GEMV


# ------------------------------------------------------------------------------------------------ GEMV
@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("m,k,b", [(16, 256, 1), (16, 256, 3), (48, 1024, 8), (33, 2048, 5), (256, 4096, 2), (7, 8192, 1)])
def test_gemv_parity(gu, name, t, m, k, b):
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=m + k)
    x = _x(b + k, b, k)
    y = ops.mul_mat(gu.qtensor(t, w, m, k), gu.to_dev(x), path=ops.PATH_GEMV).cpu().numpy()
    yo = R.o_mul_mat(t, w, x, m, k)
    e = R.rel_l2(y, yo); gu.report(test="gemv", type=name, m=m, k=k, b=b, rel_l2=e)
    assert np.isfinite(y).all() and e < TOL_GEMV


@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("m,k", [(5, 256), (130, 4096), (64, 16384), (9, 512)])
def test_fused_decode_equals_two_kernel_path(gu, name, t, m, k):
    """B=1 ggml_cdna4_mul_mat quantizes the activation row inside the GEMV launch; it must reproduce the
    quantize-then-GEMV pair bit for bit (same quantizer body, same dot bodies, same reduction order)"""
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=3 * m + k)
    x = _x(k - m, 1, k, "normal")
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y_fused = ops.mul_mat(a, xd).cpu().numpy()
    y_two = ops.mul_mat_prepared(a, ops.PreparedAct(t, xd, path=ops.PATH_GEMV)).cpu().numpy()
    assert np.array_equal(y_fused.view(np.uint32), y_two.view(np.uint32))
    e = R.rel_l2(y_fused, R.o_mul_mat(t, w, x, m, k)); gu.report(test="gemv_fused", type=name, m=m, k=k, rel_l2=e)
    assert e < TOL_GEMV


@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("m,k,b", [(130, 4096, 2), (300, 2048, 3), (64, 14336, 8), (4100, 1024, 5), (9, 512, 7)])
def test_small_batch_decode_is_one_launch_and_equals_the_two_kernel_path(gu, name, t, m, k, b):
    """2..8 activation rows: ggml_cdna4_mul_mat quantizes all of them inside ONE GEMV launch (k_gemv_q_fused<.., NB>, columns read from
    LDS; 3 rows run the 4-column form whose padding column repeats the last row; K = 14336 x 8 rows = 130 KB of LDS); bit for bit the
    quantize-then-k_gemv_q pair, and the oracle to 1e-5.  A strided activation view (row stride > K) goes through the same launch."""
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=3 * m + k + b)
    x = _x(abs(k - m) + b, b, k, "normal")
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y_one = ops.mul_mat(a, xd).cpu().numpy()
    y_two = ops.mul_mat_prepared(a, ops.PreparedAct(t, xd, path=ops.PATH_GEMV)).cpu().numpy()
    if b >= 5 and k % 256 == 0:
        # since round 4 five and more rows take the int8 matrix-core kernel in AUTO (7.9 vs 11.2 us at 4096^2, profiles/r04/batch_sweep.txt): the same integer block
        # sums, the fp32 scale products in another association — not bit for bit, but to fp32 re-association
        assert R.rel_l2(y_one, y_two) < 2e-6
    else:
        assert np.array_equal(y_one.view(np.uint32), y_two.view(np.uint32))
    e = R.rel_l2(y_one, R.o_mul_mat(t, w, x, m, k)); gu.report(test="gemv_small_batch", type=name, m=m, k=k, b=b, rel_l2=e)
    assert e < TOL_GEMV
    import torch
    wide = torch.zeros((b, k + 64), dtype=torch.float32, device="cuda")
    wide[:, :k] = xd
    y_view = ops.mul_mat(a, wide[:, :k]).cpu().numpy()
    assert np.array_equal(y_view.view(np.uint32), y_one.view(np.uint32))


@pytest.mark.parametrize("name,t", WT)
def test_gemv_many_columns_and_32_block_k(gu, name, t):
    """B > 8 through the GEMV path (column groups), and K = one block for the 32-block formats"""
    from ggml_amd import ops
    k = 32 if t in (R.Q4_0, R.Q8_0) else 256
    m, b = 20, 19
    w = R.random_weights(t, m, k, seed=3)
    x = _x(9, b, k)
    y = ops.mul_mat(gu.qtensor(t, w, m, k), gu.to_dev(x), path=ops.PATH_GEMV).cpu().numpy()
    assert R.rel_l2(y, R.o_mul_mat(t, w, x, m, k)) < TOL_GEMV


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(16, 256, 9), (128, 512, 64), (200, 1024, 100), (130, 768, 33), (512, 2048, 128), (64, 4096, 16)]


@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("m,k,b", GEMM_SHAPES)
def test_gemm_parity_auto(gu, name, t, m, k, b):
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=m * 3 + k)
    x = _x(b * 7 + k, b, k)
    # asymmetric data: a transposed C-write or a wrong k permutation cannot pass
    y = ops.mul_mat(gu.qtensor(t, w, m, k), gu.to_dev(x), path=ops.PATH_GEMM).cpu().numpy()
    yo = R.o_mul_mat(t, w, x, m, k)
    e = R.rel_l2(y, yo); ee = R.rel_l2(y, R.o_mul_mat(t, w, x, m, k, exact=True))
    gu.report(test="gemm_auto", type=name, m=m, k=k, b=b, rel_l2=e, rel_l2_exact=ee)
    assert np.isfinite(y).all() and e < TOL_GEMM


@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("variant", [6, 7, 663, 2071, 4119])
@pytest.mark.parametrize("splitk", [1, 2])
def test_gemm_variants(gu, name, t, variant, splitk):
    """the explicit kernels that remain behind gemm_variant: 6 / 7 the per-lane-load k_gemm_q without / with LDS-staged weights (128-wide tile), 663 k_gemm_kq_w8, 2071 _w8p, 4119 _w12 — x split-K
    (the 64-wide forms and k_gemm_kq_pipe went in round 5: bits 1 and 3 are ignored)"""
    from ggml_amd import ops
    m, k, b = 260, 1024, 150
    w = R.random_weights(t, m, k, seed=11)
    x = _x(12, b, k)
    y = ops.mul_mat(gu.qtensor(t, w, m, k), gu.to_dev(x), path=ops.PATH_GEMM, gemm_variant=variant, splitk=splitk).cpu().numpy()
    e = R.rel_l2(y, R.o_mul_mat(t, w, x, m, k))
    gu.report(test="gemm_variant", type=name, variant=variant, splitk=splitk, rel_l2=e)
    assert np.isfinite(y).all() and e < TOL_GEMM


@pytest.mark.parametrize("name,t", [(n, t) for n, t in WT if n in ("q4_0", "q8_0", "q6_K")])
@pytest.mark.parametrize("m,k,b", [(260, 1024, 150), (129, 768, 65), (300, 2048, 200), (4096, 4096, 512)])
def test_repacked_formats_match_per_lane_kernel(gu, name, t, m, k, b):
    """Q4_0 / Q8_0 / Q6_K at prefill batch sizes run on the LDS pipeline: with >= 3 superblocks of K per work-group the
    loader waves of k_gemm_kq_w12 re-lay the original blocks while staging them (shapes 2-4 here), shallower K goes through
    the per-call re-layout into scratch (shape 1); explicit variant 6 forces the older per-lane-load kernel on the original
    bytes.  Same arithmetic per weight (fp16 d*(q-off)), different fp32 summation order only."""
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=5 * m + k)
    x = _x(m + b, b, k, "normal" if m < 1000 else "uniform")
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y_new = ops.mul_mat(a, xd, path=ops.PATH_GEMM).cpu().numpy()
    y_old = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=6, splitk=1).cpu().numpy()
    e = R.rel_l2(y_new, y_old); gu.report(test="repack_vs_per_lane", type=name, m=m, k=k, b=b, rel_l2=e)
    assert np.isfinite(y_new).all() and e < 2e-6
    if m < 1000:
        assert R.rel_l2(y_new, R.o_mul_mat(t, w, x, m, k)) < TOL_GEMM
    # the scratch copy is rebuilt per call: a second matrix of the same shape must not see the first one's weights
    w2 = R.random_weights(t, m, k, seed=77)
    y2 = ops.mul_mat(gu.qtensor(t, w2, m, k), xd, path=ops.PATH_GEMM).cpu().numpy()
    y2_old = ops.mul_mat(gu.qtensor(t, w2, m, k), xd, path=ops.PATH_GEMM, gemm_variant=6, splitk=1).cpu().numpy()
    assert R.rel_l2(y2, y2_old) < 2e-6


T64, T64_128, T64_256 = 8192 | 7, 8192 | 16384 | 7, 8192 | 32768 | 7      # k_gemm_kq_t64: tile rows auto / 128 / 256


@pytest.mark.parametrize("variant", [T64_128, T64_256])
@pytest.mark.parametrize("m,k,b,splitk", [(128, 256, 128, 1), (300, 1536, 200, 1), (256, 2048, 128, 2), (513, 1024, 129, 2), (256, 1792, 128, 2),
                                          (700, 768, 90, 1), (1024, 4096, 512, 0), (4096, 4096, 512, 0), (4096, 2816, 512, 0)])
def test_gemm_64x128_wave_tile_kernel(gu, m, k, b, splitk, variant):
    """variant bit 13: k_gemm_kq_t64 (64(m) x 128(b) wave tiles, headers loaded per lane, reader-reset exchange flags) in both
    tile heights; ragged M / B edges, one-superblock K ranges, even and uneven hand-off splits, repeated launches on one
    scratch (a flag that its reader failed to reset would show as a stale sum)"""
    from ggml_amd import ops
    t = R.Q4_K
    w = R.random_weights(t, m, k, seed=m + k + b)
    a = gu.qtensor(t, w, m, k)
    for it in range(3):
        x = _x(m * 2 + b + it, b, k)
        xd = gu.to_dev(x)
        y = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=variant, splitk=splitk).cpu().numpy()
        yd = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=4 | 1 | 2, splitk=1).cpu().numpy()      # the 4-wave pipelined kernel: same per-weight arithmetic
        e = R.rel_l2(y, yd)
        assert np.isfinite(y).all() and e < 2e-6, (it, e)
        if m < 1100 and it == 0:
            eo = R.rel_l2(y, R.o_mul_mat(t, w, x, m, k)); gu.report(test="gemm_t64", variant=variant, m=m, k=k, b=b, splitk=splitk, rel_l2=eo)
            assert eo < TOL_GEMM
        y2 = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=variant, splitk=splitk).cpu().numpy()
        assert np.array_equal(y, y2)                                # deterministic


def test_gemm_auto_picks_the_large_tile_kernel_on_huge_grids(gu):
    """256 x 256 tiles that fill the chip (the C5-like regime): the auto path is k_gemm_r8 (round 4; 32 x 256 wave tiles, no K split here) —
    bit-identical to asking for it explicitly, within 2e-6 of k_gemm_kq_t64's 256-row form (same per-weight arithmetic, another summation order over k)
    and within tolerance of the oracle on a row sample"""
    from ggml_amd import ops
    t, m, k, b = R.Q4_K, 16384, 256, 1024
    w = R.random_weights(t, m, k, seed=3)
    x = _x(8, b, k)
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y = ops.mul_mat(a, xd).cpu().numpy()
    yx = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=(1 << 28) | (1 << 26), splitk=0).cpu().numpy()
    assert np.array_equal(y, yx)
    assert R.rel_l2(y, ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=T64_256, splitk=1).cpu().numpy()) < 2e-6
    rows = np.random.default_rng(0).choice(m, 64, replace=False)
    rs = R.row_size(t, k)
    wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
    e = R.rel_l2(y[:, rows], R.o_mul_mat(t, wsub, x, 64, k)); gu.report(test="gemm_t64_auto", m=m, k=k, b=b, rel_l2=e)
    assert np.isfinite(y).all() and e < TOL_GEMM


def test_gemm_matches_gemv_statistically(gu):
    """same inputs through both kernel families: they implement the same reference semantics"""
    from ggml_amd import ops
    t, m, k, b = R.Q4_K, 512, 4096, 8
    w = R.random_weights(t, m, k, seed=21)
    x = _x(22, b, k)
    a = gu.qtensor(t, w, m, k)
    xd = gu.to_dev(np.concatenate([x, x]))
    yv = ops.mul_mat(a, xd[:8], path=ops.PATH_GEMV).cpu().numpy()
    ym = ops.mul_mat(a, xd, path=ops.PATH_GEMM).cpu().numpy()
    assert R.rel_l2(ym[:8], yv) < TOL_GEMM
    assert np.array_equal(ym[:8], ym[8:])       # deterministic: default path uses no split-K atomics


# ------------------------------------------------------------------------------------------------ full size
@pytest.mark.parametrize("name,t", WT)
def test_full_size_decode_config(gu, name, t):
    """BASELINE configs[1]: [4096x4096]·[4096x1] — whole output against the oracle"""
    from ggml_amd import ops
    m = k = 4096
    w = R.random_weights(t, m, k, seed=1234)
    x = _x(4321, 1, k)
    y = ops.mul_mat(gu.qtensor(t, w, m, k), gu.to_dev(x)).cpu().numpy()
    e = R.rel_l2(y, R.o_mul_mat(t, w, x, m, k)); gu.report(test="full_decode", type=name, rel_l2=e)
    assert e < TOL_GEMV


@pytest.mark.parametrize("name,t", [(n, t) for n, t in WT if n in ("q4_K", "q5_K", "q6_K", "q4_0")])
@pytest.mark.parametrize("m,k", [(4096, 14336), (2048, 8192), (14336, 4096)])
def test_decode_configurations_of_long_rows_and_tall_matrices(gu, name, t, m, k):
    """the launcher's decode configurations beyond 4096^2 (gemv_q.hip: launch_fused — 16 waves x 1 row where that is one work-group per CU (round 5), 8 x 1 / 8 x 2 on
    taller matrices; rows of several rounds of 64 units: the register sets trade places every round): bit for bit the quantize-then-GEMV pair, whole output against the
    oracle (pin the configurations the route table chooses at 4096 x 14336 and 14336 x 4096)"""
    from ggml_amd import ops
    w = R.random_weights(t, m, k, seed=m + 7 * k)
    x = _x(k + m, 1, k, "normal")
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y_fused = ops.mul_mat(a, xd).cpu().numpy()
    y_two = ops.mul_mat_prepared(a, ops.PreparedAct(t, xd, path=ops.PATH_GEMV)).cpu().numpy()
    assert np.array_equal(y_fused.view(np.uint32), y_two.view(np.uint32))
    e = R.rel_l2(y_fused, R.o_mul_mat(t, w, x, m, k)); gu.report(test="gemv_fused_long", type=name, m=m, k=k, rel_l2=e)
    assert e < TOL_GEMV


@pytest.mark.parametrize("m,k,b", [(4096, 4096, 512), (4096, 11008, 512)])
def test_full_size_prefill_config(gu, m, k, b):
    """BASELINE headline / configs[2] shapes (K = 11008 = 43 whole Q4_K superblocks — an ODD count, so the auto route runs it
    without a K split today; the timing harnesses of round 1 used 10752 = 42):
    a 96-row sample of weight rows against the oracle + size-independent properties on the full output."""
    from ggml_amd import ops
    t = R.Q4_K
    w = R.random_weights(t, m, k, seed=1234)
    x = _x(4321, b, k)
    a = gu.qtensor(t, w, m, k)
    xd = gu.to_dev(x)
    y = ops.mul_mat(a, xd).cpu().numpy()
    assert np.isfinite(y).all()
    rows = np.random.default_rng(0).choice(m, 96, replace=False)
    rs = R.row_size(t, k)
    wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
    e = R.rel_l2(y[:, rows], R.o_mul_mat(t, wsub, x, 96, k)); gu.report(test="full_prefill", m=m, k=k, b=b, rel_l2=e)
    assert e < TOL_GEMM
    # property 1: permuting weight rows permutes output columns bit-exactly (row independence)
    perm = np.random.default_rng(1).permutation(m)
    wp = w.reshape(m, rs)[perm].reshape(-1)
    yp = ops.mul_mat(gu.qtensor(t, wp, m, k), xd).cpu().numpy()
    assert np.array_equal(yp, y[:, perm])
    # property 2: the result for an activation row does not depend on its batch neighbours — bit for bit while the K split is the same
    # (explicit split-K 2 on both sides), and to fp32 re-association otherwise: a 128-row batch is a quarter of the tiles and takes the
    # deep K split (8 partial sums in fixed order instead of 2)
    sk = dict(path=ops.PATH_GEMM, gemm_variant=8192 | 16384 | 7, splitk=2)
    assert np.array_equal(ops.mul_mat(a, xd[64:192].contiguous(), **sk).cpu().numpy(), ops.mul_mat(a, xd, **sk).cpu().numpy()[64:192])
    y2 = ops.mul_mat(a, xd[64:192].contiguous()).cpu().numpy()
    assert R.rel_l2(y2, y[64:192]) < 2e-6
    # property 3: row-sharded evaluation (the multi-GPU partition) concatenates to the full result: bit-identical with the same K split,
    # to re-association on the auto route (a quarter of the rows is a quarter of the tiles)
    ysh = np.concatenate([ops.mul_mat(a.rows(lo, lo + m // 4), xd, **sk).cpu().numpy() for lo in range(0, m, m // 4)], axis=1)
    assert np.array_equal(ysh, ops.mul_mat(a, xd, **sk).cpu().numpy())
    ysh = np.concatenate([ops.mul_mat(a.rows(lo, lo + m // 4), xd).cpu().numpy() for lo in range(0, m, m // 4)], axis=1)
    assert R.rel_l2(ysh, y) < 2e-6


@pytest.mark.parametrize("name,t", [("q4_K", R.Q4_K), ("q5_K", R.Q5_K), ("q6_K", R.Q6_K), ("q4_0", R.Q4_0), ("q8_0", R.Q8_0)])
@pytest.mark.parametrize("m,k,b", [(48, 1024, 2), (100, 512, 5), (512, 2048, 8), (37, 768, 9), (256, 4096, 16), (130, 1024, 17), (64, 2304, 33), (3072, 768, 64), (4096, 4096, 64),
                                   (4096, 14336, 8), (4096, 14336, 40)])
def test_small_batches_on_the_int8_matrix_cores(gu, name, t, m, k, b):
    """9 .. 32 activation rows (and 3 .. 8 over large matrices) of a Q4_K MUL_MAT take k_mmq_q4_K (mmq_i8.hip: v_mfma_i32_16x16x32_i8 on the Q8_K-quantized activations — the integer
    block dots of ggml_vec_dot_q4_K_q8_K): within the GEMV bar of the oracle (the fp16 GEMM these sizes used to take above 8 rows sits at 3e-4),
    equal to the v_dot4 GEMV units to fp32 summation order, deterministic, weight-row counts that are no multiple of 16, more rows than one
    16-column group."""
    from ggml_amd import ops
    if t != R.Q4_K and m * k > (1 << 24) and b != 8:
        pytest.skip("the largest shapes: Q4_K, and the 8-row case for the others")
    w = R.random_weights(t, m, k, seed=m + k + b) if m * k <= (1 << 24) else R.random_block_bytes(t, m, k, np.random.default_rng(m + k + b))
    x = _x(b + 3 * k, b, k)
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y = ops.mul_mat(a, xd).cpu().numpy()
    assert np.isfinite(y).all()
    rows = np.arange(m) if m <= 512 else np.random.default_rng(0).choice(m, 64, replace=False)
    rs = R.row_size(t, k)
    wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
    e = R.rel_l2(y[:, rows], R.o_mul_mat(t, wsub, x, len(rows), k)); gu.report(test="mmq_i8", type=name, m=m, k=k, b=b, rel_l2=e)
    int8_route = b <= 32 or (m * k <= (1 << 24) and t != R.Q6_K)      # capi.hip: use_mmq (GEMV or int8 MFMA up to 32 rows, 64 over small matrices — Q6_K: 32 —; above that the fp16 GEMM)
    assert e < (TOL_GEMV if int8_route else TOL_GEMM)
    assert np.array_equal(y, ops.mul_mat(a, xd).cpu().numpy())
    if b <= 16:
        assert R.rel_l2(y, ops.mul_mat(a, xd, path=ops.PATH_GEMV).cpu().numpy()) < 2e-6
    # the tail rides in the kernel's store: bit-identical to the product followed by ADD(bias) -> ADD(residual)
    if 9 <= b <= 32 and t == R.Q4_K:
        import ctypes as C
        from ggml_amd import native
        L = native.lib()
        bias = gu.to_dev(np.random.default_rng(1).standard_normal(m).astype(np.float32)); res = gu.to_dev(np.random.default_rng(2).standard_normal((b, m)).astype(np.float32))
        yt = ops.mul_mat(a, xd); yf = torch.empty_like(yt)
        ws = torch.empty(max(L.ggml_cdna4_mul_mat_workspace_size(int(t), k, b), 256), dtype=torch.uint8, device="cuda")
        native.check(L.ggml_cdna4_mul_mat_fused(int(t), a.data.data_ptr(), a.row_bytes, xd.data_ptr(), k, yf.data_ptr(), m, m, k, b, bias.data_ptr(), 0, res.data_ptr(), m,
                                                ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(yf, (yt + bias) + res)


def test_full_size_c5_config(gu):
    """BASELINE configs[4] on one GPU: Q4_K [32768 x 8192] . [8192 x 512] — the only BASELINE shape that takes the 256-row tile instantiation
    (k_gemm_kq_t64<Q4_K, 256>, no K split).  A 64-row sample of weight rows against the oracle, determinism, and the row-shard property of the
    8-GPU partition: the eight 4096-row shards (each a headline-sized product on the 128-row kernel with the split-K hand-off) concatenate to the
    full result to fp32 re-association.  Weights: random valid block bytes (the reference quantizer would take minutes for 268 M weights)."""
    from ggml_amd import ops
    t, m, k, b = R.Q4_K, 32768, 8192, 512
    w = R.random_block_bytes(t, m, k, np.random.default_rng(77))
    x = _x(78, b, k)
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y = ops.mul_mat(a, xd)
    assert torch_equal(y, ops.mul_mat(a, xd))
    yh = y.cpu().numpy()
    assert np.isfinite(yh).all()
    rows = np.random.default_rng(0).choice(m, 64, replace=False)
    rs = R.row_size(t, k)
    wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
    e = R.rel_l2(yh[:, rows], R.o_mul_mat(t, wsub, x, 64, k)); gu.report(test="full_prefill_c5", m=m, k=k, b=b, rel_l2=e)
    assert e < TOL_GEMM
    for r in (0, 5):                                                       # two of the eight row shards of the 8-GPU partition
        ysh = ops.mul_mat(a.rows(r * 4096, (r + 1) * 4096), xd).cpu().numpy()
        assert R.rel_l2(ysh, yh[:, r * 4096:(r + 1) * 4096]) < 2e-6


@pytest.mark.parametrize("m,k,b", [(4096, 4096, 512), (4096, 11008, 512)])
def test_headline_shape_against_the_oracle_full_matrix(gu, m, k, b):
    """the headline product, Q4_K [4096 x 4096] . [4096 x 512], ALL 2,097,152 outputs against the oracle's MUL_MAT (its
    OpenMP build finishes in seconds), on the PRESCRIBED inputs bench.py times (reference-quantized mt19937(1234) weights through oracle/_ref/synth_data when
    present, random valid blocks otherwise): rel-L2 over the whole matrix and the worst single output column, through AUTO (k_gemm_kq_t64, hand-off split) and
    through the shared-device route (ticketed split).  Round 5: the same for BASELINE configs[2], [4096 x 11008] . [11008 x 512] — the one
    BASELINE shape with an odd superblock count (43: the split in two is 22 + 21)."""
    import bench as B
    from ggml_amd import ops, native
    t = R.Q4_K
    w, x, how = B.prescribed(t, m, k, 0, m, b)
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    want = R.o_mul_mat(t, w, x, m, k)
    L = native.lib()
    for shared in (0, 1):
        old = L.ggml_cdna4_set_shared_device(shared)
        try:
            y = ops.mul_mat(a, xd).cpu().numpy()
        finally:
            L.ggml_cdna4_set_shared_device(old)
        e = R.rel_l2(y, want)
        col = float(np.max(np.linalg.norm(y - want, axis=0) / np.maximum(np.linalg.norm(want, axis=0), 1e-30)))
        gu.report(test="headline_full_matrix", m=m, k=k, b=b, data=how, shared_device=shared, rel_l2=e, worst_column_rel_l2=col)
        assert np.isfinite(y).all() and e < TOL_GEMM and col < 2 * TOL_GEMM, (shared, e, col)


def test_c5_shape_on_reference_quantized_rows(gu):
    """item 7(c), second half: BASELINE configs[4] [32768 x 8192] . [8192 x 512] with weight rows the REFERENCE quantized (synth_data slices the prescribed
    mt19937(1234) matrix by rows): 16 slices of 256 rows spread over the matrix are multiplied INSIDE the full-size launch (the other rows random valid
    blocks, so the launch is the real C5 grid on k_gemm_r8) and all 4096 of them are compared with the oracle."""
    import bench as B
    from ggml_amd import ops
    if not os.path.exists(os.path.join(R.REF_DIR, "synth_data")):
        pytest.skip("oracle/_ref/synth_data not built")
    t, m, k, b = R.Q4_K, 32768, 8192, 512
    rs = R.row_size(t, k)
    w = R.random_block_bytes(t, m, k, np.random.default_rng(5)).copy()
    x = None; rows = []
    for i in range(16):
        lo = i * 2048 + 512
        ws, xs, how = B.prescribed(t, m, k, lo, lo + 256, b)
        assert how == "prescribed"
        w[lo * rs:(lo + 256) * rs] = ws
        x = xs; rows.extend(range(lo, lo + 256))
    a, xd = gu.qtensor(t, w, m, k), gu.to_dev(x)
    y = ops.mul_mat(a, xd).cpu().numpy()
    rows = np.array(rows)
    wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
    e = R.rel_l2(y[:, rows], R.o_mul_mat(t, wsub, x, len(rows), k)); gu.report(test="c5_reference_quantized_rows", rows=len(rows), rel_l2=e)
    assert np.isfinite(y).all() and e < TOL_GEMM


def torch_equal(a, b):
    import torch
    return bool(torch.equal(a, b))


# ------------------------------------------------------------------------------------------------ MUL_MAT_ID
@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("n_expert,n_used,n_b_is_one,n_tok", [(4, 1, False, 1), (4, 2, False, 32), (8, 4, True, 32), (8, 2, False, 3)])
def test_mul_mat_id_parity(gu, name, t, n_expert, n_used, n_b_is_one, n_tok):
    """shapes of tests/test-backend-ops.cpp:4089-4119 (m=512, k=256)"""
    from ggml_amd import ops
    m, k = 512, 256
    rng = np.random.default_rng(n_expert * 10 + n_used)
    w = R.random_weights(t, n_expert * m, k, seed=5)
    n_b = 1 if n_b_is_one else n_used
    xb = rng.uniform(-1, 1, (n_tok, n_b, k)).astype(np.float32)
    ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    y = ops.mul_mat_id(gu.qtensor(t, w, n_expert * m, k), gu.to_dev(xb), gu.to_dev(ids), n_expert=n_expert).cpu().numpy()
    yo = R.o_mul_mat_id(t, w, xb, ids, m, k, n_expert)
    e = R.rel_l2(y, yo); gu.report(test="mul_mat_id", type=name, rel_l2=e)
    grouped = n_tok * n_used > 32                                   # capi.hip: more than 32 (token, slot) rows take the grouped fp16 GEMM (all five formats since round 3)
    assert e < (TOL_GEMM if grouped else TOL_GEMV)


@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("n_expert,n_used,n_b_is_one", [(8, 2, False), (8, 4, True), (4, 1, False)])
def test_mul_mat_id_single_token_is_one_fused_launch(gu, name, t, n_expert, n_used, n_b_is_one):
    """n_tok = 1 (mixture-of-experts decode) quantizes the activation row inside the GEMV launch; the multi-token call takes
    quantize + GEMV.  Token 0 of a two-token call must therefore equal the single-token call bit for bit."""
    from ggml_amd import ops
    m, k = 520, 1024
    rng = np.random.default_rng(n_expert + 100 * n_used)
    w = R.random_weights(t, n_expert * m, k, seed=9)
    n_b = 1 if n_b_is_one else n_used
    xb = (rng.standard_normal((2, n_b, k)) * 2).astype(np.float32)
    ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(2)]).astype(np.int32)
    a = gu.qtensor(t, w, n_expert * m, k)
    y2 = ops.mul_mat_id(a, gu.to_dev(xb), gu.to_dev(ids), n_expert=n_expert).cpu().numpy()
    y1 = ops.mul_mat_id(a, gu.to_dev(xb[:1]), gu.to_dev(ids[:1]), n_expert=n_expert).cpu().numpy()
    assert np.array_equal(y1.view(np.uint32), y2[:1].view(np.uint32))
    e = R.rel_l2(y1, R.o_mul_mat_id(t, w, xb[:1], ids[:1], m, k, n_expert)); gu.report(test="mul_mat_id_fused", type=name, rel_l2=e)
    assert e < TOL_GEMV


@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("n_expert,n_used,n_tok,m,k", [(8, 2, 64, 4096, 4096), (8, 2, 96, 512, 512), (16, 2, 40, 256, 256), (64, 4, 24, 256, 512)])
def test_mul_mat_id_few_rows_per_expert_takes_the_integer_path(gu, name, t, n_expert, n_used, n_tok, m, k):
    """more than 32 (token, slot) rows, at most 32 per expert on average (a short MoE prompt: 64 tokens x 2 of 8 experts of 4096^2): k_mmq_* over 32-row chunks of
    the expert-sorted image (VERDICT r3 "missing 4") for Q5_K / Q6_K / Q4_0 / Q8_0 — the CPU's own integer block dots: the GEMV bar (1e-5) against the oracle's MUL_MAT_ID, deterministic,
    and the time of the call beside the padded fp16 grouped GEMM it replaces goes to the report (headline-sized experts only)."""
    from ggml_amd import ops
    if m >= 4096 and t not in (R.Q4_K, R.Q6_K, R.Q8_0):
        pytest.skip("headline-sized experts: three formats")
    rng = np.random.default_rng(n_expert + n_tok)
    w = R.random_weights(t, n_expert * m, k, seed=6)
    xb = rng.uniform(-1, 1, (n_tok, n_used, k)).astype(np.float32)
    ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    a, xd, idd = gu.qtensor(t, w, n_expert * m, k), gu.to_dev(xb), gu.to_dev(ids)
    y = ops.mul_mat_id(a, xd, idd, n_expert=n_expert)
    sel = rng.choice(n_tok, min(n_tok, 16), replace=False)
    e = R.rel_l2(y.cpu().numpy()[sel], R.o_mul_mat_id(t, w, xb[sel], ids[sel], m, k, n_expert))
    assert torch.equal(y, ops.mul_mat_id(a, xd, idd, n_expert=n_expert))
    us = None
    if m >= 4096:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10): ops.mul_mat_id(a, xd, idd, n_expert=n_expert)
        e0.record()
        for _ in range(30): ops.mul_mat_id(a, xd, idd, n_expert=n_expert)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
    gu.report(test="mul_mat_id_few_rows_per_expert", type=name, n_expert=n_expert, n_used=n_used, n_tok=n_tok, m=m, k=k, rel_l2=e, us_per_call=us)
    assert e < (TOL_GEMM if t == R.Q4_K else TOL_GEMV), e           # (Q4_K stays on its grouped fp16 GEMM: 45 vs 82 us at 64 tokens of 8 x 4096^2; the others take the integer path)


@pytest.mark.parametrize("name,t", WT)
@pytest.mark.parametrize("n_expert,n_used,n_b_is_one,n_tok,m,k", [(8, 2, False, 512, 4096, 4096), (8, 2, False, 96, 512, 512), (4, 4, True, 33, 300, 768),
                                                                  (16, 2, False, 40, 256, 256), (8, 1, False, 700, 640, 1024), (3, 2, False, 200, 128, 2048)])
def test_mul_mat_id_grouped_prefill(gu, name, t, n_expert, n_used, n_b_is_one, n_tok, m, k):
    """prefill-sized MUL_MAT_ID (n_tok * n_used > 32) on all five formats.  Q4_K (round 6): the stream-k form — planner + token-order quantizer in one launch, then ONE
    persistent k_gemm_kq_sk launch over (tile, m-tile, superblock) units that gathers its activation rows; Q5_K / Q6_K / Q4_0 / Q8_0: the (token, slot) rows counting-sorted
    by expert, quantized through the sort (gather) and multiplied by k_gemm_q<.., IDS> (per-lane loads of the original blocks); ragged per-expert counts, experts that receive no row, an expert id out of range (its
    slot must stay untouched), b broadcast over the slots (n_b = 1), repeated calls on one workspace.  Against the oracle's MUL_MAT_ID (per-expert
    mul_mat, ggml-cpu.c:7648-7781) with the GEMM tolerance."""
    from ggml_amd import ops
    if m >= 4096 and t != R.Q4_K and t != R.Q6_K:
        pytest.skip("headline-sized experts: Q4_K and one per-lane-load format")
    rng = np.random.default_rng(n_expert * 10 + n_used + n_tok)
    w = R.random_weights(t, n_expert * m, k, seed=5)
    n_b = 1 if n_b_is_one else n_used
    xb = rng.uniform(-1, 1, (n_tok, n_b, k)).astype(np.float32)
    ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    if n_expert == 16:
        ids[ids == 5] = 6                                            # expert 5 gets no row at all
    a = gu.qtensor(t, w, n_expert * m, k)
    y = ops.mul_mat_id(a, gu.to_dev(xb), gu.to_dev(ids), n_expert=n_expert).cpu().numpy()
    if m * n_tok <= 300000:
        yo = R.o_mul_mat_id(t, w, xb, ids, m, k, n_expert)
        e = R.rel_l2(y, yo)
    else:                                                            # headline-sized experts: a sample of the (token, slot) pairs
        sel = rng.choice(n_tok, 24, replace=False)
        yo = R.o_mul_mat_id(t, w, xb[sel], ids[sel], m, k, n_expert)
        e = R.rel_l2(y[sel], yo)
    gu.report(test="mul_mat_id_grouped", n_expert=n_expert, n_used=n_used, n_tok=n_tok, m=m, k=k, rel_l2=e)
    assert np.isfinite(y).all() and e < TOL_GEMM
    y2 = ops.mul_mat_id(a, gu.to_dev(xb), gu.to_dev(ids), n_expert=n_expert).cpu().numpy()
    assert np.array_equal(y, y2)                                     # the sort order is not deterministic, the result is
    # an out-of-range id leaves its slot untouched
    ids_bad = ids.copy(); ids_bad[3, 0] = n_expert + 7
    out = torch.full((n_tok, n_used, m), -77.0, dtype=torch.float32, device="cuda")
    L = __import__("ggml_amd.native", fromlist=["lib"]).lib()
    nws = L.ggml_cdna4_mul_mat_id_workspace_size(int(t), k, n_expert, n_used, n_b, n_tok)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    xd, idd = gu.to_dev(xb), gu.to_dev(ids_bad)
    rc = L.ggml_cdna4_mul_mat_id(int(t), a.data.data_ptr(), a.row_bytes, m * a.row_bytes, xd.data_ptr(), k, n_b * k, idd.data_ptr(), n_used,
                                 out.data_ptr(), m, n_used * m, m, k, n_expert, n_used, n_b, n_tok, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.ggml_cdna4_last_error().decode()
    torch.cuda.synchronize()
    ob = out.cpu().numpy()
    # (the other rows: equal to the clean run up to fp32 summation order — one row fewer moves the spans of the stream-k partition, i.e. where a tile's K range is cut and summed)
    assert (ob[3, 0] == -77.0).all() and R.rel_l2(ob[3, 1:], y[3, 1:]) < 2e-6 and R.rel_l2(ob[4], y[4]) < 2e-6 and R.rel_l2(ob[5:], y[5:]) < 2e-6

def test_shared_device_mode_never_spins_and_agrees(gu):
    """GGML_CDNA4_SHARED_DEVICE=1 (ggml_cdna4_set_shared_device): the AUTO routes choose no split-K exchange that waits for a co-resident partner —
    k_gemm_kq_t64 takes the ticketed split in two at M = 4096 (uneven for K = 11008), the large-grid shapes drop k_gemm_r8's split, the 128 x 128-tile
    kernels run unsplit — and every product stays within the fp32 summation-order distance of the default route.  Timings of both
    modes go to the report."""
    import subprocess, sys, json
    code = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import refutil as R
from ggml_amd import ops, native
L = native.lib()
out = {}
for (t, m, k, b) in ((R.Q4_K, 4096, 4096, 512), (R.Q4_K, 4096, 11008, 512), (R.Q4_K, 16384, 4096, 512), (R.Q4_0, 4096, 4096, 512), (R.Q6_K, 2048, 4096, 128)):
    w = R.random_weights(t, m, k, seed=m + k)
    a = ops.QTensor.from_host_bytes(t, k, m, w)
    xd = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (b, k)).astype(np.float32)).cuda()
    ys, us = [], []
    for shared in (0, 1, 0):
        L.ggml_cdna4_set_shared_device(shared)
        y = ops.mul_mat(a, xd); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20): ops.mul_mat(a, xd)
        e0.record()
        for _ in range(50): ops.mul_mat(a, xd)
        e1.record(); e1.synchronize()
        ys.append(y.cpu().numpy()); us.append(e0.elapsed_time(e1) * 20.0)
    out["%%d %%dx%%dx%%d" %% (t, m, k, b)] = {"rel_l2_shared_vs_default": R.rel_l2(ys[1], ys[0]), "default_again_identical": bool(np.array_equal(ys[0], ys[2])),
                                          "finite": bool(np.isfinite(ys[1]).all()), "us_default": round(us[0], 2), "us_shared": round(us[1], 2)}
print(json.dumps(out))
""" % (R.ROOT, os.path.join(R.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    gu.report(test="shared_device_mode", **{k.replace(" ", "_"): v for k, v in rec.items()})
    for k, v in rec.items():
        assert v["finite"] and v["default_again_identical"] and v["rel_l2_shared_vs_default"] < 2e-6, (k, v)
