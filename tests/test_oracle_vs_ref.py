"""Pins oracle/ggml_oracle.c against the UNMODIFIED reference compiled in place (oracle/_ref, built by
oracle/ref.mk): bit-exact for quantize_row / dequantize_row, rel-L2 <= 2e-6 for MUL_MAT (only the
order of the final fp32 adds differs between the portable and the AVX2 vec_dot bodies).
Skipped where oracle/_ref is not built; the committed fixtures (test_oracle_golden.py) cover that case."""
import numpy as np
import pytest
import refutil as R

pytestmark = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built (needs /root/reference)")

WT = [R.Q4_0, R.Q8_0, R.Q4_K, R.Q5_K, R.Q6_K] + list(R.ORACLE_ONLY_TYPES.values())     # the HIP formats + the oracle-only ones


def _data(seed, shape, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, shape).astype(np.float32)
    if kind == "normal":
        return (rng.standard_normal(shape) * 3).astype(np.float32)
    if kind == "cos":      # tests/test-quantize-fns.cpp:31-35
        n = int(np.prod(shape))
        return (0.1 + 2 * np.cos(np.arange(n, dtype=np.float32) + seed)).astype(np.float32).reshape(shape)
    if kind == "ties":     # values on exact .5 grid points to exercise rounding rules
        return (rng.integers(-254, 255, shape) / 2.0).astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("t", WT)
def test_dequantize_bit_exact(t):
    k = 1024
    w = R.r_quantize(t, _data(1, (8, k), "uniform"))
    assert np.array_equal(R.o_dequantize(t, w, k).view(np.uint32), R.r_dequantize(t, w, k).view(np.uint32))


@pytest.mark.parametrize("kind", ["uniform", "normal", "cos", "ties"])
@pytest.mark.parametrize("t", [R.Q4_0, R.Q4_K, R.Q4_1])        # one weight type per activation format: Q8_0, Q8_K, Q8_1
def test_activation_quantize_bit_exact(t, kind):
    x = _data(7, (16, 2048), kind)
    x[3, 256:512] = 0.0          # an all-zero block
    x[5, 7] = -x[5, :256].__abs__().max() * 2   # negative max
    a = R.o_quantize_act(t, x)
    b = R.r_quantize_act(t, x)
    if t == R.Q4_K:              # all-zero Q8_K block: reference leaves bsums unwritten (src/ggml-quants.c:2492-2497)
        a = a.reshape(16, -1, 292).copy(); b = b.reshape(16, -1, 292).copy()
        zero = np.all(a[:, :, 4:260] == 0, axis=2) & np.all(a[:, :, 0:4] == 0, axis=2)
        a[zero, 260:] = 0; b[zero, 260:] = 0
    assert np.array_equal(a, b)


def test_q8_0_ref_and_q4_0_ref_bit_exact():
    import ctypes as C
    base, _ = R.ref()
    x = _data(11, (1, 4096), "normal")
    for name, t, sym in [("q8_0_ref", R.Q8_0, "quantize_row_q8_0_ref"), ("q4_0_ref", R.Q4_0, "quantize_row_q4_0_ref"), ("q4_1_ref", R.Q4_1, "quantize_row_q4_1_ref"),
                         ("q5_0_ref", R.Q5_0, "quantize_row_q5_0_ref"), ("q5_1_ref", R.Q5_1, "quantize_row_q5_1_ref")]:
        out = np.zeros(R.row_size(t, 4096), np.uint8)
        getattr(base, sym)(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(4096))
        assert np.array_equal(out, R.o_quantize_row(name, x)), name


@pytest.mark.parametrize("t", WT)
def test_mul_mat_matches_reference(t):
    m, k, b = 48, 1024, 5
    w = R.r_quantize(t, _data(21, (m, k), "uniform"))
    x = _data(22, (b, k), "uniform")
    yo, yr = R.o_mul_mat(t, w, x, m, k), R.r_mul_mat(t, w, x, m, k)
    assert R.rel_l2(yo, yr) < 2e-6
    # and the CPU path really is ~4e-3 away from the exact product
    ye = R.o_mul_mat(t, w, x, m, k, exact=True)
    assert 5e-4 < R.rel_l2(yr, ye) < 2e-2


def test_fp16_roundtrip_all_values():
    o = R.oracle()
    h = np.arange(65536, dtype=np.uint16)
    f = h.view(np.float16).astype(np.float32)
    for v in list(range(0, 65536, 97)) + [0, 1, 0x3ff, 0x400, 0x7bff, 0x8001, 0xfbff]:
        got = o.oracle_fp16_to_fp32(v)
        if np.isnan(f[v]):
            assert np.isnan(got)
        else:
            assert got == f[v]
            assert o.oracle_fp32_to_fp16(float(f[v])) == v
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.standard_normal(3000) * 10.0 ** rng.integers(-8, 5, 3000), [65504, 65519.9, 65520, 1e-8, 5.96e-8, 2.98e-8]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    for x, w in zip(xs, want):
        assert o.oracle_fp32_to_fp16(float(x)) == w, (x, w)


@pytest.mark.parametrize("D,nh,nhk,nq,nkv,max_bias,softcap,use_mask", [(64, 4, 4, 5, 96, 0.0, 0.0, True), (128, 4, 2, 35, 200, 8.0, 0.0, True), (128, 2, 2, 3, 64, 0.0, 10.0, True),
                                                                        (256, 2, 1, 33, 130, 0.0, 0.0, False), (80, 3, 3, 7, 64, 0.0, 0.0, True)])
def test_flash_attn_ext_oracle_vs_reference(D, nh, nhk, nq, nkv, max_bias, softcap, use_mask):
    """oracle_flash_attn_ext_f16 against ggml_flash_attn_ext on the reference CPU backend: same fp16 Q / accumulator roundings, only the
    summation order inside the K.Q dot differs (an fp16 accumulator turns that into occasional 1-ulp(fp16) flips): rel-L2 <= 2e-4.
    And the reference itself is 1-3e-3 from a float64 evaluation — the number the GPU bars in test_gpu_widening.py are set against."""
    import refops as O
    rng = np.random.default_rng(D + nq)
    q = rng.uniform(-1, 1, (1, nh, nq, D)).astype(np.float32)
    k = rng.uniform(-1, 1, (1, nhk, nkv, D)).astype(np.float16); v = rng.uniform(-1, 1, (1, nhk, nkv, D)).astype(np.float16)
    m = rng.uniform(-1, 1, ((nq + 63) // 64 * 64, nkv)).astype(np.float16) if use_mask else None
    if use_mask:
        m[:, ::7] = -np.inf
    scale = float(1 / np.sqrt(D))
    yo = R.o_flash_attn_ext(q, k, v, m, scale, max_bias, softcap)
    yr = O.flash_attn_ext(q, k, v, m, scale, max_bias, softcap)
    assert R.rel_l2(yo, yr) < 2e-4
    ye = R.exact_flash_attn_ext(q, k, v, m, scale, max_bias, softcap)
    assert 2e-4 < R.rel_l2(yr, ye) < 6e-3
