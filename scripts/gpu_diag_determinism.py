"""hardware diagnosis: repeated identical MUL_MAT calls must be bit-identical.  Per (type, shape, splitk): how many of N repeats differ from the first,
and how many are WRONG against the oracle on 48 sampled weight rows (rel-L2 > 1e-3) — written for the non-deterministic IQ4_XS 4096x4096x512 result of
round 3's first hardware sessions (two-part Q6_K GEMM, K' = 8192, split-K hand-off of k_gemm_kq_w12)."""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R
from ggml_amd import ops

def run(name, t, m, k, b, splitk=0, n=12):
    w = R.random_weights(t, m, k, seed=5 * m + k)
    x = np.random.default_rng(b * 7 + k).uniform(-1, 1, (b, k)).astype(np.float32)
    a = ops.QTensor.from_host_bytes(t, k, m, w); xd = torch.from_numpy(x).cuda()
    rows = np.random.default_rng(0).choice(m, 48, replace=False); rs = R.row_size(t, k)
    want = R.o_mul_mat(t, np.concatenate([w[r * rs:(r + 1) * rs] for r in rows]), x, len(rows), k)
    ys = [ops.mul_mat(a, xd, splitk=splitk).cpu().numpy() for _ in range(n)]
    errs = [R.rel_l2(y[:, rows], want) for y in ys]
    diff = [int((y != ys[0]).sum()) for y in ys]
    blocks = sorted({(int(i) // 128, int(j) // 128) for y in ys[1:3] for i, j in np.argwhere(y != ys[0])[:4000]})[:12]
    print(json.dumps({"type": name, "shape": [m, k, b], "splitk": splitk, "wrong_vs_oracle": int(sum(e > 1e-3 for e in errs)), "max_err": float(max(errs)), "first_err": float(errs[0]),
                      "differ_from_first": int(sum(d > 0 for d in diff)), "n_diff": diff[:6], "tiles(b128,m128)": blocks}), flush=True)

if __name__ == "__main__":
    T = {"iq4_xs": R.IQ4_XS, "q2_K": R.Q2_K, "q6_K": R.Q6_K}
    for name, m, k, b, sk in [("iq4_xs", 4096, 4096, 512, 0), ("iq4_xs", 4096, 4096, 512, 1), ("iq4_xs", 4096, 2048, 512, 0), ("iq4_xs", 2048, 4096, 512, 0), ("iq4_xs", 4096, 4096, 128, 0),
                              ("iq4_xs", 4096, 4096, 1024, 0), ("iq4_xs", 8192, 4096, 512, 0), ("q2_K", 4096, 4096, 512, 0), ("q2_K", 8192, 4096, 512, 0)]:
        run(name, T[name], m, k, b, sk)
