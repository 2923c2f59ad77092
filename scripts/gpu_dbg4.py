import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import refutil as R
from ggml_amd import native, ops
t, m, k, b = R.Q4_K, 16384, 256, 1024
w = R.random_weights(t, m, k, seed=3)
x = np.random.default_rng(8).uniform(-1, 1, (b, k)).astype(np.float32)
a = ops.QTensor.from_host_bytes(t, k, m, w); xd = torch.from_numpy(x).cuda()
T256 = 8192 | 32768 | 7
ys = {}
for name, kw in (("auto", {}), ("t256_a", dict(path=ops.PATH_GEMM, gemm_variant=T256, splitk=1)), ("t256_b", dict(path=ops.PATH_GEMM, gemm_variant=T256, splitk=1)),
                 ("auto2", {}), ("ref4w", dict(path=ops.PATH_GEMM, gemm_variant=7, splitk=1)), ("w12", dict(path=ops.PATH_GEMM, gemm_variant=4119, splitk=1))):
    ys[name] = ops.mul_mat(a, xd, **kw).cpu().numpy()
for n in ys:
    d = ys[n] - ys["ref4w"]
    bad = np.argwhere(np.abs(d) > 1e-3 * np.abs(ys["ref4w"]).max())
    print(n, "rel_l2 vs 4-wave", R.rel_l2(ys[n], ys["ref4w"]), "equal to auto:", np.array_equal(ys[n], ys["auto"]), "n bad", len(bad), "first bad (b,m)", bad[:3].tolist())
