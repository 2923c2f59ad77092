"""decode loop for profiling: the one-launch Q4_K GEMV (k_gemv_q_fused) streaming N rotating weight matrices from HBM (cold: 64 x 33 MB > every cache).
    python scripts/decode_loop.py [M K steps]     prints us per step (HIP events)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R
from ggml_amd import ops

M, K, steps = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 14336, 400)
nmat = max(4, min(64, int(3e9 // (M * K * 0.5625))))
base = R.random_block_bytes(R.Q4_K, M, K, np.random.default_rng(1))
mats = [ops.QTensor.from_host_bytes(R.Q4_K, K, M, base) for _ in range(nmat)]
x = torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, (1, K)).astype(np.float32)).cuda()
y = torch.empty((1, M), dtype=torch.float32, device="cuda")
for i in range(50):
    ops.mul_mat(mats[i % nmat], x, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps):
    ops.mul_mat(mats[i % nmat], x, out=y)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / steps
nbytes = M * K * 0.5625 + 4 * K + 4 * M
print("decode %dx%d: %.2f us/step, %.0f GB/s (%.3f of 8 TB/s), %d rotating matrices" % (M, K, us, nbytes / us / 1e3, nbytes / us / 8e6, nmat))
