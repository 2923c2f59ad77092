"""Reference formulas shared by the emulator harness (emul_check.py): the pair-interleaved k order of the fp16 activation
image and the fp16 (s, c) constants of a Q4_K / Q5_K 64-k group, restated in numpy (test infrastructure)."""
import numpy as np

PERM = [0, 2, 1, 3]                                   # the fp16 activation image stores (k0, k2, k1, k3) within every 4


def f16(x):
    return np.float16(x)


def k4_scale_min(scales12, jj):
    """get_scale_min_k4, /root/reference/src/ggml-quants.c:631-638"""
    q = scales12
    if jj < 4:
        return q[jj] & 63, q[jj + 4] & 63
    return (q[jj + 4] & 0xF) | ((q[jj - 4] >> 6) << 4), (q[jj + 4] >> 4) | ((q[jj] >> 6) << 4)


def table_entry(block, G, zero=8.0):
    """(sl, cl, sh, ch) of 64-k group G of one superblock: the arithmetic of Raw<Q4_K / Q5_K>::scales() / tab_store (zero = 8 / 16)"""
    d = np.float32(block[0:2].view(np.float16)[0]); dmin = np.float32(block[2:4].view(np.float16)[0])
    s0, m0 = k4_scale_min(block[4:16], 2 * G); s1, m1 = k4_scale_min(block[4:16], 2 * G + 1)
    sl, sh = f16(d * np.float32(s0)), f16(d * np.float32(s1))
    cl = f16(np.float32(zero) * np.float32(sl) - dmin * np.float32(m0)); ch = f16(np.float32(zero) * np.float32(sh) - dmin * np.float32(m1))
    return sl, cl, sh, ch
