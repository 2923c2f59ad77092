"""Random shapes through the whole-library emulation (lib_emul_check): every accepted weight type, AUTO route, against the oracle.
    python tools/emul/lib_emul_fuzz.py [n_cases [seed]]        prints one line per case, FAIL lines for anything outside the bars"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import lib_emul_check as L  # noqa: E402
import refutil as R  # noqa: E402

TYPES = [R.Q4_0, R.Q4_1, R.Q5_0, R.Q5_1, R.Q8_0, R.Q2_K, R.Q3_K, R.Q4_K, R.Q5_K, R.Q6_K, R.IQ4_NL, R.IQ4_XS]


def main(n, seed):
    rng = np.random.default_rng(seed)
    fails = 0
    for i in range(n):
        t = int(rng.choice(TYPES))
        blk = R.BLCK[t]
        k = int(blk * rng.integers(1, (int(rng.choice([2304, 2304, 4608])) // blk) + 1))
        m = int(rng.choice([1, 2, 3, 5, 16, 17, 31, 33, 64, 100, 129, 200, 257, 400, 513]))
        b = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 31, 33, 64, 100, 129, 200, 300]))
        cus = int(rng.choice([256, 256, 8, 1]))                     # the CU count the launchers size their grids / K splits for
        if rng.random() < 0.3:
            ne, nu = int(rng.choice([2, 4, 8])), int(rng.choice([1, 2]))
            nb, nt = int(rng.choice([1, nu])), int(rng.choice([1, 1, 2, 5, 20, 40, 70]))
            t0 = time.time()
            try:
                e = L.mul_mat_id(t, min(m, 64), k, ne, nu, nb, nt, seed=i, cus=cus, timeout=240)
                ok = e is None or e < (1e-3 if (t == R.Q4_K and nt * nu > 32) else 1e-5)        # Q4_K with more than 32 (token, slot) rows: the grouped MFMA GEMM
            except Exception as ex:  # noqa: BLE001
                e, ok = repr(ex)[-200:], False
            print("%s id  type %2d m %3d k %4d experts %d used %d n_b %d tok %d  %s  %.0fs" % ("ok  " if ok else "FAIL", t, min(m, 64), k, ne, nu, nb, nt, e, time.time() - t0), flush=True)
        else:
            t0 = time.time()
            try:
                r = L.mul_mat(t, m, k, b, seed=i, cus=cus, timeout=240)
                e = None if r is None else r[0]
                ok = e is None or e < (1e-5 if b <= 8 else 1e-3)
            except Exception as ex:  # noqa: BLE001
                e, ok = repr(ex)[-200:], False
            print("%s mm  type %2d m %3d k %4d b %3d cus %3d  %s  %.0fs" % ("ok  " if ok else "FAIL", t, m, k, b, cus, e, time.time() - t0), flush=True)
        fails += not ok
    print("%d cases, %d failures" % (n, fails))
    return fails


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
