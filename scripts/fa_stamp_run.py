import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ggml_amd import ops
D, H = int(sys.argv[1]), 32
n = 4096
g = torch.Generator().manual_seed(1)
q = (torch.rand((1, H, n, D), generator=g) * 2 - 1).cuda()
k = (torch.rand((1, H, n, D), generator=g) * 2 - 1).half().cuda()
v = (torch.rand((1, H, n, D), generator=g) * 2 - 1).half().cuda()
m = (torch.rand((n, n), generator=g) * 2 - 1).half().cuda()
if os.environ.get("FA_LOOP_CAUSAL"):
    m = torch.triu(torch.full((n, n), float("-inf"), dtype=torch.float16, device="cuda"), diagonal=1)
for _ in range(int(os.environ.get("FA_STAMP_CALLS", "3"))):
    ops.flash_attn_ext(q, k, v, m if len(sys.argv) < 3 else None, float(1 / np.sqrt(D)))
torch.cuda.synchronize()
