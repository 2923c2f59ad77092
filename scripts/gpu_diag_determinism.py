"""hardware diagnosis: repeated identical MUL_MAT calls must be bit-identical.  Reports, per (type, shape), how many of N repeats differ from the
first, and where (tile coordinates of the differing elements) — written for the one non-deterministic IQ4_XS 4096x4096x512 result of round 3's
first hardware session (two-part Q6_K GEMM, K' = 8192, split-K hand-off of k_gemm_kq_w12)."""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R
from ggml_amd import ops

def run(name, t, m, k, b, n=30):
    w = R.random_weights(t, m, k, seed=5 * m + k)
    x = np.random.default_rng(b * 7 + k).uniform(-1, 1, (b, k)).astype(np.float32)
    a = ops.QTensor.from_host_bytes(t, k, m, w); xd = torch.from_numpy(x).cuda()
    y0 = ops.mul_mat(a, xd).cpu().numpy()
    bad = []
    for i in range(n):
        y = ops.mul_mat(a, xd).cpu().numpy()
        d = np.argwhere(y != y0)
        if d.size:
            rows, cols = np.unique(d[:, 0] // 64), np.unique(d[:, 1] // 32)
            bad.append({"rep": i, "n": int(len(d)), "max_abs": float(np.abs(y - y0).max()), "b_blocks64": rows.tolist()[:16], "m_blocks32": cols.tolist()[:16],
                        "nan": int(np.isnan(y).sum())})
    print(json.dumps({"type": name, "shape": [m, k, b], "repeats": n, "differing": len(bad), "detail": bad[:4]}), flush=True)

if __name__ == "__main__":
    for name, t, m, k, b in [("iq4_xs", R.IQ4_XS, 4096, 4096, 512), ("q2_K", R.Q2_K, 4096, 4096, 512), ("q6_K", R.Q6_K, 4096, 8192, 512), ("q6_K", R.Q6_K, 4096, 4096, 512),
                             ("q8_0", R.Q8_0, 4096, 8192, 512), ("q4_K", R.Q4_K, 4096, 4096, 512), ("q5_K", R.Q5_K, 4096, 4096, 512)]:
        run(name, t, m, k, b)
