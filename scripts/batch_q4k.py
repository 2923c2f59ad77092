"""Q4_K small-batch sweep on DEVICE time (HIP-graph replay of 40 calls): us per ggml_cdna4_mul_mat call for 1 .. 64 activation rows at 4096^2 and 4096 x 14336, under the
environment of the process (routing knobs are read once) — printed as one JSON line with a tag"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
from ggml_amd import native, ops
native.lib(); dev = torch.device("cuda", 0)
out = {"tag": os.environ.get("AB_TAG", "")}
shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("BATCH_SHAPES", "4096x4096,4096x14336").split(",")]
rows = [int(v) for v in os.environ.get("BATCH_ROWS", "1,2,3,4,6,8,12,16,24,32,48,64").split(",")]
for (m, k) in shapes:
    a = ops.QTensor.from_host_bytes(12, k, m, B.synth_blocks(12, m, k, 7), device=dev)
    row = {}
    for b in rows:
        x = torch.from_numpy(np.random.default_rng(b).uniform(-1, 1, (b, k)).astype(np.float32)).to(dev)
        y = torch.empty((b, m), dtype=torch.float32, device=dev)
        row[b] = round(B.graph_us(dev, lambda: ops.mul_mat(a, x, out=y), 40), 2)
    out["%dx%d" % (m, k)] = row
print(json.dumps(out), flush=True)
