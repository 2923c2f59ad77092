"""The CPU oracle against the committed fixtures produced by the unmodified reference
(tests/golden/make_golden.py).  Runs anywhere (no /root/reference, no GPU needed)."""
import os
import numpy as np
import pytest
import refutil as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mul_mat_small.npz"))
GID = np.load(os.path.join(os.path.dirname(__file__), "golden", "mul_mat_id_small.npz"))
GM = np.load(os.path.join(os.path.dirname(__file__), "golden", "mul_mat_more_formats.npz"))       # Q4_1 / Q5_0 / Q5_1 / Q2_K / Q3_K (oracle only so far)
M, K, B = int(G["M"]), int(G["K"]), int(G["B"])


@pytest.mark.parametrize("name", list(R.QUANT_TYPES))
def test_dequantize_golden(name):
    t = R.QUANT_TYPES[name]
    assert np.array_equal(R.o_dequantize(t, G[name + "_w"], K).view(np.uint32), G[name + "_deq"].view(np.uint32))


@pytest.mark.parametrize("name", list(R.QUANT_TYPES))
def test_activation_quantize_golden(name):
    t = R.QUANT_TYPES[name]
    a, b = R.o_quantize_act(t, G[name + "_x"]), G[name + "_act"]
    if t in (R.Q4_K, R.Q5_K, R.Q6_K):   # all-zero Q8_K blocks: the reference leaves bsums unwritten
        a = a.reshape(B, -1, 292).copy(); b = b.reshape(B, -1, 292).copy()
        z = np.all(b[:, :, 0:260] == 0, axis=2)
        a[z, 260:] = 0; b[z, 260:] = 0
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name", list(R.QUANT_TYPES))
def test_mul_mat_golden(name):
    t = R.QUANT_TYPES[name]
    y = R.o_mul_mat(t, G[name + "_w"], G[name + "_x"], M, K)
    assert R.rel_l2(y, G[name + "_y"]) < 2e-6


@pytest.mark.parametrize("name", list(R.QUANT_TYPES))
def test_mul_mat_id_golden(name):
    t = R.QUANT_TYPES[name]
    y = R.o_mul_mat_id(t, GID[name + "_w"], GID[name + "_x"], GID["ids"], int(GID["M"]), int(GID["K"]), int(GID["n_expert"]))
    assert R.rel_l2(y, GID[name + "_y"]) < 2e-6


def test_mul_mat_id_broadcast_b():
    # b.ne[1] == 1: every slot reads the same activation row (src/ggml-cpu/ggml-cpu.c:7752)
    t = R.Q4_K
    x = GID["q4_K_x"][:, :1, :]
    y = R.o_mul_mat_id(t, GID["q4_K_w"], x, GID["ids"], int(GID["M"]), int(GID["K"]), int(GID["n_expert"]))
    y2 = R.o_mul_mat_id(t, GID["q4_K_w"], np.repeat(x, int(GID["n_used"]), axis=1), GID["ids"], int(GID["M"]), int(GID["K"]), int(GID["n_expert"]))
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("name", list(R.ORACLE_ONLY_TYPES))
def test_more_formats_golden(name):
    """the formats no HIP kernel takes yet: dequantize and the activation quantizer bit-exact,
    MUL_MAT to summation order, against fixtures from the unmodified reference"""
    t = R.ORACLE_ONLY_TYPES[name]
    assert np.array_equal(R.o_dequantize(t, GM[name + "_w"], K).view(np.uint32), GM[name + "_deq"].view(np.uint32))
    a, b = R.o_quantize_act(t, GM[name + "_x"]), GM[name + "_act"]
    if R.act_type(t) == R.Q8_K:             # all-zero Q8_K blocks: the reference leaves bsums unwritten
        a = a.reshape(B, -1, 292).copy(); b = b.reshape(B, -1, 292).copy()
        z = np.all(b[:, :, 0:260] == 0, axis=2)
        a[z, 260:] = 0; b[z, 260:] = 0
    assert np.array_equal(a, b)
    assert R.rel_l2(R.o_mul_mat(t, GM[name + "_w"], GM[name + "_x"], M, K), GM[name + "_y"]) < 2e-6
