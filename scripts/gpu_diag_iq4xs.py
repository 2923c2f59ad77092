"""hardware diagnosis, IQ4_XS prefill (two-part Q6_K GEMM) non-determinism: is it the GEMM on this data, or the in-library route (conversion into
scratch, doubled image)?  The same converted bytes (made on the CPU by tools/emul/convert_emul, i.e. the conversion kernel's own source) are fed as a
NATIVE Q6_K tensor with K' = 2K against [x x]: the identical product through the plain Q6_K route."""
import os, sys, json, subprocess, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R
from ggml_amd import ops

def reps(fn, n=10):
    ys = [fn().cpu().numpy() for _ in range(n)]
    d = [int((y != ys[0]).sum()) for y in ys]
    mt = np.zeros(ys[0].shape[1] // 128, int)
    for y in ys[1:]:
        mt += (y != ys[0]).reshape(y.shape[0], -1, 128).any(axis=(0, 2))
    return ys[0], d, mt

m, k, b = 4096, 4096, 512
w = R.random_weights(R.IQ4_XS, m, k, seed=5 * m + k)
x = np.random.default_rng(b * 7 + k).uniform(-1, 1, (b, k)).astype(np.float32)
xd = torch.from_numpy(x).cuda()
a = ops.QTensor.from_host_bytes(R.IQ4_XS, k, m, w)
y0, d, mt = reps(lambda: ops.mul_mat(a, xd))
print(json.dumps({"route": "iq4_xs in-library", "n_diff_vs_first": d, "m_tiles_ever_differing": int((mt > 0).sum()), "of": int(mt.size)}), flush=True)
exe = os.path.join(ROOT, "tools", "emul", "convert_emul")
with tempfile.TemporaryDirectory() as td:
    w.tofile(os.path.join(td, "w.bin"))
    r = subprocess.run([exe, str(R.IQ4_XS), str(m), str(k), os.path.join(td, "w.bin"), os.path.join(td, "o.bin")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    cw = np.fromfile(os.path.join(td, "o.bin"), np.uint8)
a6 = ops.QTensor.from_host_bytes(R.Q6_K, 2 * k, m, cw)
x2 = torch.from_numpy(np.concatenate([x, x], axis=1)).cuda()
y6, d6, mt6 = reps(lambda: ops.mul_mat(a6, x2))
print(json.dumps({"route": "same bytes as native Q6_K, K' = 8192, [x x]", "n_diff_vs_first": d6, "m_tiles_ever_differing": int((mt6 > 0).sum()), "equal_to_in_library_first": bool(np.array_equal(y6, y0)),
                  "rel_l2_vs_in_library_first": float(np.linalg.norm(y6 - y0) / np.linalg.norm(y0))}), flush=True)
# the in-library route again AFTER the native one (same scratch, warmed)
y1, d1, mt1 = reps(lambda: ops.mul_mat(a, xd))
print(json.dumps({"route": "iq4_xs in-library, again", "n_diff_vs_first": d1, "equal_native_first": bool(np.array_equal(y1, y6))}), flush=True)
