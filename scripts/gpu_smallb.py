"""µs per MUL_MAT call (activation quantize included, HIP events) over the batch sizes between decode and prefill, Q4_K"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import refutil as R
from ggml_amd import ops
import bench
t = R.Q4_K
for (m, k) in ((4096, 4096), (4096, 14336)):
    w = R.random_weights(t, m, k, seed=3)
    a = ops.QTensor.from_host_bytes(t, k, m, w)
    row = []
    for b in (1, 2, 4, 8, 9, 16, 32, 48, 64, 65, 96, 128, 256, 512):
        x = torch.from_numpy(np.random.default_rng(b).uniform(-1, 1, (b, k)).astype(np.float32)).cuda()
        out = torch.empty((b, m), dtype=torch.float32, device="cuda")
        f = lambda: ops.mul_mat(a, x, out=out)
        us = bench.events_us(f, 200, 20)
        row.append("B=%d: %.1f" % (b, us))
    print("%dx%d  " % (m, k) + "  ".join(row), flush=True)
