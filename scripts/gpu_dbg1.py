import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import refutil as R
from ggml_amd import native
import test_gpu_cabi_ops as T
L = native.lib()
rows, k = 37, 1024
x = T._data("uniform", (rows, k), 11); x[3, 32:64] = 0; x[4, 5] = -7.5; x[4, 9] = 7.5
xd = T._dev(x)
out = torch.zeros(rows * R.row_size(R.Q4_0, k), dtype=torch.uint8, device="cuda")
rc = L.ggml_cdna4_op_cpy(C.byref(T._desc(xd, R.F32)), C.byref(T._qdesc(out, R.Q4_0, k, rows)), 1, T._st())
torch.cuda.synchronize()
got = out.cpu().numpy().reshape(-1, 18)
want = np.concatenate([R.o_quantize_row("q4_0_ref", x[i]) for i in range(rows)]).reshape(-1, 18)
bad = np.argwhere(got != want)
print("rc", rc, "mismatching bytes", len(bad), "of", got.size, "blocks with a mismatch", len(set(bad[:, 0])))
for b, j in bad[:8]:
    xb = x.reshape(-1, 32)[b]
    print("block", b, "byte", j, "got", got[b, j], "want", want[b, j], "d got/want", got[b, :2].view(np.float16), want[b, :2].view(np.float16), "amax idx", np.argmax(np.abs(xb)), xb[np.argmax(np.abs(xb))])
