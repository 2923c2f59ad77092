"""Q5_K on k_gemm_r8 vs its 128x128-tile kernel k_gemm_kq_w8p, same box, kernel only (activations prepared), + a row sample against the oracle"""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench as B
import refutil as R
dev = torch.device("cuda", 0)
from ggml_amd import native; native.lib()
R8, W8P = (1 << 28) | (1 << 26), 2048 | 16 | 7
out = {}
for (m, k, b) in ((32768, 8192, 512), (32768, 4096, 512), (16384, 8192, 1024), (16384, 4096, 1024)):
    w = B.synth_blocks(13, m, k, 3)
    x = np.random.default_rng(1).uniform(-1, 1, (b, k)).astype(np.float32)
    h = B.Hot(dev, 13, w, m, k, x)
    h.prepare()
    row = {}
    for name, v in (("w8p", W8P), ("r8", R8), ("w8p_again", W8P), ("r8_again", R8), ("auto", 0)):
        h.variant = v
        row[name] = round(B.events_us(h.gemm_only, 30, 10), 2)
    h.variant = 0; h.step(); torch.cuda.synchronize()
    y = h.y.cpu().numpy()
    rows = np.random.default_rng(0).choice(m, 32, replace=False)
    rs = R.row_size(13, k)
    wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
    row["rel_l2_vs_oracle_32_rows"] = R.rel_l2(y[:, rows], R.o_mul_mat(13, wsub, x, 32, k))
    out["q5_K %dx%dx%d" % (m, k, b)] = row
    print("q5_K %dx%dx%d" % (m, k, b), row, flush=True)
print(json.dumps(out))
