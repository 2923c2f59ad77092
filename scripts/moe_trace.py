"""per-segment timing of k_gemm_kq_sk (the instrumented twin in tools/microbench/libcdna4_kernels_abl.so; CDNA4_KERNELS_LIB points at it): bench.moe_row's shape, one call
traced — for every span its segments' prologue / k loop / epilogue in us, and the launch's critical path"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CDNA4_KERNELS_LIB", os.path.join(ROOT, "tools", "microbench", "libcdna4_kernels_abl.so"))
import numpy as np, torch
import bench
from ggml_amd import native, ops
L = native.lib()
dev = torch.device("cuda", 0)
n_expert, n_used, n_tok, m, k = 8, 2, 512, 4096, 4096
w, _, how = bench.prescribed(12, n_expert * m, k, 0, n_expert * m, 1)
a = ops.QTensor.from_host_bytes(12, k, n_expert * m, w, device=dev)
rng = np.random.default_rng(7)
xb = torch.from_numpy(rng.uniform(-1, 1, (n_tok, n_used, k)).astype(np.float32)).to(dev)
ids = torch.from_numpy(np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)).to(dev)
for _ in range(5): ops.mul_mat_id(a, xb, ids, n_expert=n_expert)
G = torch.cuda.get_device_properties(0).multi_processor_count
tr = torch.zeros(G * 64, dtype=torch.int64, device=dev)
L.ggml_cdna4_debug_trace(tr.data_ptr())
ops.mul_mat_id(a, xb, ids, n_expert=n_expert)
torch.cuda.synchronize()
L.ggml_cdna4_debug_trace(None)
t = tr.cpu().numpy().reshape(G, 8, 8)
t0 = min(int(t[w, 0, 0]) for w in range(G) if t[w, 0, 0])
rows = []
for w in range(G):
    for s in range(8):
        if not t[w, s, 3]: continue
        st, b0, lp, en, nsb, nfrag, nparts = [int(v) for v in t[w, s, :7]]
        rows.append(dict(w=w, s=s, start=(st - t0) / 100.0, pro=(b0 - st) / 100.0, loop=(lp - b0) / 100.0, epi=(en - lp) / 100.0, nsb=nsb, nfrag=nfrag, parts=nparts, end=(en - t0) / 100.0))
import collections
print("# spans %d, segments %d, launch critical path %.2f us (first segment start -> last segment end)" % (G, len(rows), max(r["end"] for r in rows)))
for key, sel in (("whole tiles", lambda r: r["parts"] == 1), ("cut tiles", lambda r: r["parts"] > 1)):
    rr = [r for r in rows if sel(r)]
    if rr: print("%s: n %d  prologue %.2f  loop/superblock %.3f  epilogue %.2f us (means)" % (key, len(rr), np.mean([r["pro"] for r in rr]), np.mean([r["loop"] / r["nsb"] for r in rr]), np.mean([r["epi"] for r in rr])))
for f in (1, 2, 3, 4):
    rr = [r for r in rows if r["nfrag"] == f]
    if rr: print("fragments %d: n %d  loop per superblock %.3f us" % (f, len(rr), np.mean([r["loop"] / r["nsb"] for r in rr])))
ends = sorted(max(r["end"] for r in rows if r["w"] == w) for w in set(r["w"] for r in rows))
print("span end times: min %.1f  median %.1f  max %.1f us; first-segment start: min %.2f max %.2f" % (ends[0], ends[len(ends) // 2], ends[-1], min(r["start"] for r in rows if r["s"] == 0), max(r["start"] for r in rows if r["s"] == 0)))
for r in rows[:24]: print(json.dumps(r))
