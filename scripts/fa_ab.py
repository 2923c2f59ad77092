"""FLASH_ATTN_EXT prefill A/B on one box: the pipelined kernel (CDNA4_FA_PIPE = 8 / 4 / 2 waves per work-group, or the launcher's own choice) against the older kernels
(CDNA4_FA_PIPE=0), settings alternating, through the C-ABI.  Prints us per call, TFLOP/s and the rel-L2 distance of each setting's output from the older kernels'.
    python scripts/fa_ab.py [hs ...]"""
import ctypes as C
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggml_amd import native, ops

def events_us(fn, n, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

H = 32
shapes = [(4096, 4096), (2048, 2048), (1024, 1024), (512, 512), (512, 4096), (256, 2048)]
if os.environ.get("FA_AB_SHAPES"): shapes = [tuple(int(x) for x in s_.split("x")) for s_ in os.environ["FA_AB_SHAPES"].split(",")]
settings = [("0", "older"), (None, "auto"), ("8", "pipe8"), ("4", "pipe4"), ("2", "pipe2")]
if os.environ.get("FA_AB_SETTINGS"): settings = [s_ for s_ in settings if s_[1] in os.environ["FA_AB_SETTINGS"].split(",")]
for D in ([int(a) for a in sys.argv[1:]] or [128, 64]):
    for n_q, n_kv in shapes:
        g = torch.Generator().manual_seed(1)
        q = (torch.rand((1, H, n_q, D), generator=g) * 2 - 1).cuda()
        k = (torch.rand((1, H, n_kv, D), generator=g) * 2 - 1).half().cuda()
        v = (torch.rand((1, H, n_kv, D), generator=g) * 2 - 1).half().cuda()
        mpad = int(os.environ.get("FA_AB_MASK_PAD", "0"))              # extra elements per mask row (a row stride that is no power of two)
        m = (torch.rand(((n_q + 63) // 64 * 64, n_kv + mpad), generator=g) * 2 - 1).half().cuda()[:, :n_kv]
        sc = float(1.0 / np.sqrt(D))
        o = ops.flash_attn_ext(q, k, v, m.contiguous(), sc)            # (the output buffer; the timed calls hand the strided mask to the C-ABI themselves)
        dq, dk, dv, dd = (ops._tensor_desc(t_, ty) for t_, ty in ((q, 0), (k, 1), (v, 1), (o, 0)))
        dm = ops._tensor_desc(m[None, None], 1)
        st, L = torch.cuda.current_stream().cuda_stream, native.lib()
        pm = None if os.environ.get("FA_AB_NOMASK") else C.byref(dm)
        call = lambda: native.check(L.ggml_cdna4_op_flash_attn_ext(C.byref(dq), C.byref(dk), C.byref(dv), pm, C.byref(dd), sc, 0.0, 0.0, st))
        ref, row = None, []
        for rep in range(2):
            for val, name in settings:
                if val is None: os.environ.pop("CDNA4_FA_PIPE", None)
                else: os.environ["CDNA4_FA_PIPE"] = val
                o.zero_(); call(); torch.cuda.synchronize()
                y = o.float().cpu().numpy().copy()
                if ref is None: ref = y
                us = events_us(call, 20 if n_q * n_kv >= 1 << 22 else 100)
                tf = 4.0 * H * n_q * n_kv * D / us / 1e6
                row.append("%s %.1f us %.0f TF d=%.1e" % (name, us, tf, float(np.linalg.norm(y - ref) / np.linalg.norm(ref))))
        os.environ.pop("CDNA4_FA_PIPE", None)
        print("hs%d q%d kv%d | " % (D, n_q, n_kv) + " | ".join(row), flush=True)
