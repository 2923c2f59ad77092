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
    """undo the pair-interleaved fp16 image: stored (k0,k2,k1,k3) -> natural order"""
    a = xh.reshape(xh.shape[0], -1, 4)
    return a[:, :, [0, 2, 1, 3]].reshape(xh.shape)
