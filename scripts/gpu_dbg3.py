import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import refutil as R
from ggml_amd import native, ops
L = native.lib()
t = R.Q4_K
n_expert, n_used, n_tok, m, k = 4, 2, 40, 256, 512
rng = np.random.default_rng(1)
w = R.random_weights(t, n_expert * m, k, seed=5)
xb = rng.uniform(-1, 1, (n_tok, n_used, k)).astype(np.float32)
ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
a = ops.QTensor.from_host_bytes(t, k, n_expert * m, w)
nws = L.ggml_cdna4_mul_mat_id_workspace_size(int(t), k, n_expert, n_used, n_used, n_tok)
ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
out = torch.full((n_tok, n_used, m), -77.0, dtype=torch.float32, device="cuda")
xd, idd = torch.from_numpy(xb).cuda(), torch.from_numpy(ids).cuda()
rc = L.ggml_cdna4_mul_mat_id(int(t), a.data.data_ptr(), a.row_bytes, m * a.row_bytes, xd.data_ptr(), k, n_used * k, idd.data_ptr(), n_used,
                             out.data_ptr(), m, n_used * m, m, k, n_expert, n_used, n_used, n_tok, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("rc", rc, L.ggml_cdna4_last_error().decode(), "ws bytes", nws)
R_ = (n_tok * n_used + 127) // 128 * 128 + 128 * n_expert
al = lambda x: (x + 255) & ~255
wsn = ws.cpu().numpy()
img_src = wsn[0:R_ * 4].view(np.int32); img_dst = wsn[al(R_ * 4):al(R_ * 4) + R_ * 4].view(np.int32); te = wsn[2 * al(R_ * 4):2 * al(R_ * 4) + R_ // 128 * 4].view(np.int32)
print("img_rows", R_, "tile_expert", te, "valid rows", int((img_src >= 0).sum()), "of", n_tok * n_used)
for e in range(n_expert): print(" expert", e, "count in ids", int((ids == e).sum()))
print("img_src[:10]", img_src[:10], "img_dst[:10]", img_dst[:10])
o = out.cpu().numpy()
print("untouched", int((o == -77).sum()), "nonfinite", int((~np.isfinite(o)).sum()), "of", o.size)
yo = R.o_mul_mat_id(t, w, xb, ids, m, k, n_expert)
good = np.isfinite(o) & (o != -77)
print("rel_l2 on finite", R.rel_l2(np.where(good, o, 0), np.where(good, yo, 0)))
# c5-like NaN check
from bench import prescribed, Hot
M, K, B = 8192, 8192, 512
wq, x, how = prescribed(12, M, K, 0, M, B)
for (mm, variant) in ((M, 0), (M, 8192 | 32768 | 7), (M, 4119)):
    h = Hot(torch.device("cuda", 0), 12, wq[: mm * (K // 256 * 144)], mm, K, x, variant)
    h.step(); torch.cuda.synchronize()
    y = h.y.cpu().numpy()
    print("shape", mm, K, B, "variant", variant, "nonfinite", int((~np.isfinite(y)).sum()), "absmax", float(np.nanmax(np.abs(y))))
