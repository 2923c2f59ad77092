"""FLASH_ATTN_EXT loop for profiling: prefill (n_q = n_kv, head size 128, 32 heads, causal-free random mask) through ops.flash_attn_ext.
    python scripts/fa_loop.py [n_q n_kv steps]     prints us per call (HIP events)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggml_amd import ops

n_q, n_kv, steps = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 20)
D, H = 128, 32
g = torch.Generator().manual_seed(1)
q = (torch.rand((1, H, n_q, D), generator=g) * 2 - 1).cuda()
k = (torch.rand((1, H, n_kv, D), generator=g) * 2 - 1).half().cuda()
v = (torch.rand((1, H, n_kv, D), generator=g) * 2 - 1).half().cuda()
m = (torch.rand(((n_q + 63) // 64 * 64, n_kv), generator=g) * 2 - 1).half().cuda()
if os.environ.get("FA_LOOP_CAUSAL"):
    m = torch.triu(torch.full((n_q, n_kv), float("-inf"), dtype=torch.float16, device="cuda"), diagonal=1)
scale = float(1.0 / np.sqrt(D))
for _ in range(3):
    ops.flash_attn_ext(q, k, v, m, scale)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    ops.flash_attn_ext(q, k, v, m, scale)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / steps
flops = 4.0 * n_q * n_kv * D * H
print("flash_attn_ext hs=%d heads=%d n_q=%d n_kv=%d: %.1f us/call, %.1f TFLOP/s (%.3f of 2516.6)" % (D, H, n_q, n_kv, us, flops / us / 1e6, flops / us / 1e6 / 2516.6))
