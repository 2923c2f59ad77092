"""FLASH_ATTN_EXT decode loop for profilers: one query row, 32 heads, head size 128, 32 K keys — F16 cache, the same on 8 K / V heads (grouped-query), Q8_0 and Q4_0 caches.
    python scripts/fa_decode_loop.py [steps]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggml_amd import ops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(5)
hs, nh, n_kv = 128, 32, 32768
dev = "cuda"
q = torch.from_numpy(rng.uniform(-1, 1, (1, nh, 1, hs)).astype(np.float32)).to(dev)
mk = torch.from_numpy(rng.uniform(-1, 1, (64, n_kv)).astype(np.float16)).to(dev)
sc = float(1.0 / np.sqrt(hs))
def rows(bb, nrows):
    a = rng.integers(0, 256, (nrows, hs // 32, bb), dtype=np.uint8)
    a[:, :, 0:2] = np.frombuffer(np.float16(1.0 / 64).tobytes(), dtype=np.uint8)
    return a.reshape(nrows, hs // 32 * bb)
cases = []
for nkvh in (32, 8):
    k = torch.from_numpy(rng.uniform(-1, 1, (1, nkvh, n_kv, hs)).astype(np.float16)).to(dev)
    v = torch.from_numpy(rng.uniform(-1, 1, (1, nkvh, n_kv, hs)).astype(np.float16)).to(dev)
    cases.append(("f16 kv heads %d" % nkvh, k, v, None))
for name, t, bb in (("q8_0", 8, 34), ("q4_0", 2, 18)):
    k = torch.from_numpy(rows(bb, nh * n_kv).reshape(1, nh, n_kv, hs // 32 * bb)).to(dev)
    v = torch.from_numpy(rows(bb, nh * n_kv).reshape(1, nh, n_kv, hs // 32 * bb)).to(dev)
    cases.append((name, k, v, t))
for name, k, v, t in cases:
    f = (lambda: ops.flash_attn_ext(q, k, v, mk, sc)) if t is None else (lambda: ops.flash_attn_ext(q, k, v, mk, sc, kv_type=t))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); e1.synchronize()
    print("decode %s: %.1f us per call" % (name, e0.elapsed_time(e1) * 1e3 / steps))
