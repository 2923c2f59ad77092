"""helpers shared by the -m gpu parity tests: everything goes through the C-ABI (ggml_amd.ops -> ctypes)."""
import json
import os
import numpy as np
import torch
import refutil as R
from ggml_amd import ops

REPORT = os.path.join(R.ROOT, "gpurun_out", "parity_report.jsonl")


def report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(kw) + "\n")


def qtensor(t, w_bytes, m, k):
    return ops.QTensor.from_host_bytes(t, k, m, w_bytes)


def to_dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def uninterleave(xh):
    """undo the private layout of the fp16 activation image: k-panel-major ([K/128][B][128]) and pair-interleaved
    (stored (k0,k2,k1,k3) within every 4) -> natural [B][K]"""
    B, K = xh.shape
    if K % 128 == 0:
        nat = xh.reshape(K // 128, B, 128).transpose(1, 0, 2).reshape(B, K)
    else:                                       # K not a whole number of panels: element (b,k) at ((k>>7)*B+b)*128 + (k&127)
        flat = xh.reshape(-1)
        kk = np.arange(K)
        nat = np.stack([flat[((kk >> 7) * B + b) * 128 + (kk & 127)] for b in range(B)])
    a = nat.reshape(B, -1, 4)
    return a[:, :, [0, 2, 1, 3]].reshape(B, K)
