"""stress: the 64x128-wave-tile kernels called many times on the same inputs; every call must give the same bits as the first and
agree with the 4-wave kernel to fp16-activation accuracy"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import refutil as R
from ggml_amd import ops
t = R.Q4_K
for (m, k, b, var, sk) in ((16384, 256, 1024, 8192 | 32768 | 7, 1), (16384, 1024, 1024, 8192 | 32768 | 7, 1), (4096, 4096, 512, 8192 | 16384 | 7, 2),
                           (8192, 4096, 2048, 8192 | 32768 | 7, 1), (8192, 4096, 2048, 8192 | 16384 | 7, 1)):
    w = R.random_weights(t, m, k, seed=3)
    x = np.random.default_rng(8).uniform(-1, 1, (b, k)).astype(np.float32)
    a = ops.QTensor.from_host_bytes(t, k, m, w); xd = torch.from_numpy(x).cuda()
    ref = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=7, splitk=1)
    y0 = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=var, splitk=sk).clone()
    nbad = 0
    for it in range(200):
        y = ops.mul_mat(a, xd, path=ops.PATH_GEMM, gemm_variant=var, splitk=sk)
        if not torch.equal(y, y0):
            nbad += 1
    rel = float((y0 - ref).norm() / ref.norm())
    print("%dx%dx%d variant %d splitk %d: rel vs 4-wave %.2e, %d of 200 repeats differ" % (m, k, b, var, sk, rel, nbad), flush=True)
    assert rel < 1e-5 and nbad == 0
print("STRESS OK")
