"""Random FLASH_ATTN_EXT cases through tools/emul/fattn_emul (the op's host code + the kernels from source) against a float64 evaluation.
    python tools/emul/fattn_emul_fuzz.py [n_cases [seed]]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import fattn_emul_check as F  # noqa: E402
import refutil as R  # noqa: E402


def main(n, seed):
    rng = np.random.default_rng(seed)
    fails = 0
    for i in range(n):
        D = int(rng.choice([64, 128, 256, 64, 128, 80, 96, 40]))
        nh_kv = int(rng.choice([1, 2, 3]))
        nh = nh_kv * int(rng.choice([1, 2, 4]))
        n_q = int(rng.choice([1, 1, 2, 3, 7, 31, 32, 33, 64, 65, 130]))
        n_kv = int(rng.choice([1, 5, 31, 32, 33, 64, 96, 100, 255, 256, 257, 511, 700, 1025, 2049, 4100]))
        kw = dict(D=D, n_q=n_q, n_head=nh, n_kv=n_kv, n_head_kv=nh_kv, n_batch=int(rng.choice([1, 1, 2])), mask=bool(rng.random() < 0.8), permuted=bool(rng.random() < 0.3),
                  cus=int(rng.choice([1, 2, 8, 256])), seed=i)
        if kw["mask"]:
            kw["max_bias"] = float(rng.choice([0.0, 0.0, 8.0])); kw["inf_every"] = int(rng.choice([0, 0, 3, 7]))
        if rng.random() < 0.2:
            kw["softcap"] = 10.0
        t0 = time.time()
        try:
            if rng.random() < 0.25 and D % 32 == 0:
                t = int(rng.choice([R.Q8_0, R.Q4_0, R.Q5_1]))
                r = F.run_quantized(t, D, n_q, nh, n_kv, n_head_kv=nh_kv, permuted=kw["permuted"], seed=i, cus=kw["cus"], timeout=300)
                ok = r is None or (r[0] < 5e-4 and r[1] is True)
                kw = dict(kw, kv_type=t)
            else:
                r = F.run(timeout=300, **kw)
                ok = r is None or (r[0] < 5e-4 and r[1] < 3e-2)
        except Exception as ex:  # noqa: BLE001
            r, ok = repr(ex)[-300:], False
        print("%s %s -> %s  %.0fs" % ("ok  " if ok else "FAIL", kw, r, time.time() - t0), flush=True)
        fails += not ok
    print("%d cases, %d failures" % (n, fails))
    return fails


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
