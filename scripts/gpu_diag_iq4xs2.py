"""hardware diagnosis #2 of the IQ4_XS prefill route (two-part Q6_K GEMM): WHICH weight rows differ from call to call, and is it the conversion kernel's
output or the GEMM's reading of it?  (round 3 left: whole output columns — weight rows — differ between identical calls, n_diff = columns x B.)
  A  bytes of ggml_cdna4_convert_weights (GPU, caller-owned buffer, repeated) vs the conversion kernel's source run on the CPU (tools/emul/convert_emul)
  B  native Q6_K GEMM on the CPU-converted bytes with [x x]: the control
  C  per call: GPU conversion into a caller-owned buffer, then the native Q6_K GEMM on it
  D  the in-library route: map of differing columns vs B, their byte addresses in the scratch, magnitudes
Run with CDNA4_IQ4_XS_GEMM=1 CDNA4_DIAG_CONVERT_ANY=1."""
import os, sys, json, subprocess, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R
from ggml_amd import ops, native

T = R.IQ4_XS if len(sys.argv) < 2 else getattr(R, sys.argv[1])
m, k, b = 4096, 4096, 512
L = native.lib()
w = R.random_weights(T, m, k, seed=5 * m + k)
x = np.random.default_rng(b * 7 + k).uniform(-1, 1, (b, k)).astype(np.float32)
xd = torch.from_numpy(x).cuda()
a = ops.QTensor.from_host_bytes(T, k, m, w)
exe = os.path.join(ROOT, "tools", "emul", "convert_emul")
with tempfile.TemporaryDirectory() as td:
    w.tofile(os.path.join(td, "w.bin"))
    r = subprocess.run([exe, str(int(T)), str(m), str(k), os.path.join(td, "w.bin"), os.path.join(td, "o.bin")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    cw = np.fromfile(os.path.join(td, "o.bin"), np.uint8)
rowb = cw.size // m
# A
nbytes = L.ggml_cdna4_convert_weights_size(int(T), m, k)
assert nbytes == cw.size, (nbytes, cw.size)
buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
diffs = []
for rep in range(6):
    buf.fill_(0xEE if rep % 2 else 0x11)
    native.check(L.ggml_cdna4_convert_weights(int(T), a.data.data_ptr(), a.row_bytes, m, k, buf.data_ptr(), ops._stream(buf.device)))
    g = buf.cpu().numpy()
    bad = np.nonzero(g != cw)[0]
    diffs.append(int(bad.size))
    if bad.size:
        rows = np.unique(bad // rowb)
        print(json.dumps({"A_rep": rep, "bad_bytes": int(bad.size), "bad_rows": rows[:20].tolist(), "first_bad_offsets_in_row": (bad[:16] % rowb).tolist(), "got": g[bad[:8]].tolist(), "want": cw[bad[:8]].tolist()}), flush=True)
print(json.dumps({"A": "GPU conversion vs CPU-run source", "bad_bytes_per_rep": diffs}), flush=True)
# B
a6 = ops.QTensor.from_host_bytes(R.Q6_K, 2 * k, m, cw)
x2 = torch.from_numpy(np.concatenate([x, x], axis=1)).cuda()
yB = [ops.mul_mat(a6, x2).cpu().numpy() for _ in range(6)]
print(json.dumps({"B": "native Q6_K on CPU-converted bytes", "n_diff_vs_first": [int((y != yB[0]).sum()) for y in yB], "nan": int(np.isnan(yB[0]).sum())}), flush=True)
ref = yB[0]
def colmap(y, tag):
    d = y != ref
    cols = np.nonzero(d.any(axis=0))[0]
    mag = np.abs(y - ref)[:, cols].max(axis=0) if cols.size else np.zeros(0)
    full = int((d.sum(axis=0)[cols] == y.shape[0]).sum()) if cols.size else 0
    print(json.dumps({tag: "columns (weight rows) differing from the control", "n_cols": int(cols.size), "cols_fully_different": full, "n_elems": int(d.sum()),
                      "cols": cols[:48].tolist(), "col_mod_128": (cols[:48] % 128).tolist(), "max_abs_diff": [float("%.3g" % v) for v in mag[:24]], "ref_scale": float(np.abs(ref).mean())}), flush=True)
    return cols
# C
a6g = ops.QTensor(R.Q6_K, 2 * k, m, buf)
for rep in range(6):
    native.check(L.ggml_cdna4_convert_weights(int(T), a.data.data_ptr(), a.row_bytes, m, k, buf.data_ptr(), ops._stream(buf.device)))
    y = ops.mul_mat(a6g, x2).cpu().numpy()
    colmap(y, "C_rep%d" % rep)
# D
for rep in range(6):
    y = ops.mul_mat(a, xd).cpu().numpy()
    cols = colmap(y, "D_rep%d" % rep)
# D with a device-wide sync in front of every call and an untouched stream in between
for rep in range(3):
    torch.cuda.synchronize()
    y = ops.mul_mat(a, xd); torch.cuda.synchronize(); y = y.cpu().numpy()
    colmap(y, "Dsync_rep%d" % rep)
# E: the in-library route on the slow per-lane-load kernel (an explicit variant): conversion + doubled image, another GEMM
for rep in range(3):
    y = ops.mul_mat(a, xd, gemm_variant=6).cpu().numpy()
    d = np.abs(y - ref); print(json.dumps({"E_rep%d" % rep: "in-library, k_gemm_q", "rel_l2_vs_control": float(np.linalg.norm(y - ref) / np.linalg.norm(ref)), "n_cols_gt_1e-3": int(((d.max(axis=0)) > 1e-3 * np.abs(ref).max()).sum())}), flush=True)
