"""us per ggml_cdna4_mul_mat call (quantizer launch included, HIP events, 200 calls) at a few activation-row counts.
    python scripts/batch_rows.py q6_K [M K]        CDNA4_NO_MMQ=1 in the environment gives the route without the int8 matrix-core kernel"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R
from ggml_amd import ops
name = sys.argv[1] if len(sys.argv) > 1 else "q6_K"
m, k = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 14336)
ty = {"q4_K": R.Q4_K, "q5_K": R.Q5_K, "q6_K": R.Q6_K, "q4_0": R.Q4_0, "q8_0": R.Q8_0}[name]
a = ops.QTensor.from_host_bytes(ty, k, m, R.random_block_bytes(ty, m, k, np.random.default_rng(1)))
out = []
for b in (4, 8, 16, 32, 64):
    x = torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, (b, k)).astype(np.float32)).cuda()
    y = torch.empty((b, m), dtype=torch.float32, device="cuda")
    for _ in range(20):
        ops.mul_mat(a, x, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.mul_mat(a, x, out=y)
    e1.record(); e1.synchronize()
    out.append("%d: %.2f" % (b, e0.elapsed_time(e1) * 5.0))
print("%s %dx%d%s  rows: us  |  %s" % (name, m, k, " (CDNA4_NO_MMQ)" if os.environ.get("CDNA4_NO_MMQ") else "", "   ".join(out)))
