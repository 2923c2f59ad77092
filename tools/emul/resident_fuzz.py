"""Random shapes through the whole-library emulation WITH a resident image registered (round 5: Q4_0R / Q8_0R / Q6_K8 re-layouts and the exact re-encodings), pretend CU counts
that send small grids down the large-grid kernels (k_gemm_r8 whole rounds / ragged round / split 2-4-8, k_gemm_kq_t64 128- / 256-row tiles and its splits), LDS-DMA deferred
to the counted waits on those kernels, against the oracle.
    python tools/emul/resident_fuzz.py [n_cases [seed]]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import lib_emul_check as L  # noqa: E402
import refutil as R  # noqa: E402

TYPES = [R.Q4_0, R.Q8_0, R.Q6_K, R.Q4_0, R.Q8_0, R.Q6_K, R.Q5_0, R.Q3_K, R.Q2_K, R.IQ4_XS]


def main(n, seed):
    rng = np.random.default_rng(seed)
    fails = 0
    for i in range(n):
        t = int(rng.choice(TYPES))
        k = int(256 * rng.integers(1, 10))
        m = int(rng.choice([64, 100, 256, 257, 300, 512, 513, 700, 768, 1024]))
        b = int(rng.choice([9, 33, 65, 100, 128, 129, 200, 256, 257, 300, 512, 600]))
        cus = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 256]))
        t0 = time.time()
        try:
            r = L.mul_mat(t, m, k, b, seed=i, cus=cus, timeout=900, resident=True, defer_dma=2)         # (2: deferred where every wait of the kernel is a counted one in the source — lib_emul_main.cpp)
            e = None if r is None else r[0]
            ok = e is None or e < 1e-3
        except Exception as ex:  # noqa: BLE001
            e, ok = repr(ex)[-300:], False
        print("%s type %2d m %4d k %4d b %3d cus %3d  %s  %.0fs" % ("ok  " if ok else "FAIL", t, m, k, b, cus, e, time.time() - t0), flush=True)
        fails += not ok
    print("%d cases, %d failures" % (n, fails))
    return fails


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
