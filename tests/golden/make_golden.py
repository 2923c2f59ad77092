"""Generates tests/golden/*.npz by running the UNMODIFIED reference (oracle/_ref, built by oracle/ref.mk from
/root/reference) on fixed-seed inputs.  Run from the repo root in the build container:

    make -C oracle && python tests/golden/make_golden.py

The fixtures pin (a) the CPU oracle on machines without /root/reference and (b) the HIP path on the GPU box.
Inputs follow tests/test-backend-ops.cpp:37-126 (uniform(-1,1) through ggml_quantize_chunk) and
tests/test-quantize-fns.cpp:31-35 (0.1 + 2cos(i+offset))."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refutil as R  # noqa: E402

assert R.have_ref(), "build oracle/_ref first (make -C oracle)"

M, K, B = 24, 768, 4
out = {}
for name, t in R.QUANT_TYPES.items():
    rng = np.random.default_rng(1000 + t)
    wf = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, K)).astype(np.float32)
    x[1] = (0.1 + 2 * np.cos(np.arange(K, dtype=np.float32) + t)).astype(np.float32)
    x[2, 256:512] = 0.0
    w = R.r_quantize(t, wf)
    out[name + "_w"] = w
    out[name + "_x"] = x
    out[name + "_deq"] = R.r_dequantize(t, w, K)
    out[name + "_act"] = R.r_quantize_act(t, x)
    out[name + "_y"] = R.r_mul_mat(t, w, x, M, K)
np.savez_compressed(os.path.join(HERE, "mul_mat_small.npz"), M=M, K=K, B=B, **out)

# MUL_MAT_ID fixture (shapes of tests/test-backend-ops.cpp:4089-4119, shrunk): through the reference vec_dot
n_expert, n_used, n_tok, Mi, Ki = 4, 2, 5, 16, 256
rng = np.random.default_rng(77)
ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
mm = {}
for name, t in (("q4_K", R.Q4_K), ("q8_0", R.Q8_0), ("q4_0", R.Q4_0), ("q5_K", R.Q5_K), ("q6_K", R.Q6_K)):   # (first two first: their draws predate the other three)
    w = R.r_quantize(t, rng.uniform(-1, 1, (n_expert * Mi, Ki)).astype(np.float32))
    xb = rng.uniform(-1, 1, (n_tok, n_used, Ki)).astype(np.float32)
    y = np.zeros((n_tok, n_used, Mi), np.float32)
    rs = R.row_size(t, Ki)
    for tk in range(n_tok):
        for u in range(n_used):
            e = ids[tk, u]
            y[tk, u] = R.r_mul_mat(t, w[e * Mi * rs:(e + 1) * Mi * rs], xb[tk, u:u + 1], Mi, Ki)[0]
    mm[name + "_w"], mm[name + "_x"], mm[name + "_y"] = w, xb, y
np.savez_compressed(os.path.join(HERE, "mul_mat_id_small.npz"), n_expert=n_expert, n_used=n_used, n_tok=n_tok, M=Mi, K=Ki, ids=ids, **mm)
# the widening formats (Q4_1 / Q5_0 / Q5_1 / Q2_K / Q3_K / IQ4_NL / IQ4_XS): same recipe, separate file
more = {}
for name, t in R.ORACLE_ONLY_TYPES.items():
    rng = np.random.default_rng(1000 + t)
    wf = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, K)).astype(np.float32)
    x[1] = (0.1 + 2 * np.cos(np.arange(K, dtype=np.float32) + t)).astype(np.float32)
    x[2, 256:512] = 0.0
    w = R.r_quantize(t, wf)
    more[name + "_w"], more[name + "_x"] = w, x
    more[name + "_deq"] = R.r_dequantize(t, w, K)
    more[name + "_act"] = R.r_quantize_act(t, x)
    more[name + "_y"] = R.r_mul_mat(t, w, x, M, K)
np.savez_compressed(os.path.join(HERE, "mul_mat_more_formats.npz"), M=M, K=K, B=B, **more)
print("wrote", os.listdir(HERE))
