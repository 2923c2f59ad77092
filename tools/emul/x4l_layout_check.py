"""Index-level emulation of k_gemm_q4k_x4l's data movement (ggml_amd/csrc/gemm_q_x4l.hip) on the CPU: the loader waves'
global -> LDS mapping (activation pieces with the row&15 chunk swizzle, nibble pieces with the (row>>2)&3 swizzle, the (s, c)
table), the compute waves' LDS reads (read_xa / read_w), Raw<Q4_K>::pairbits' nibble -> k order and the lane layout of
v_mfma_f32_32x32x16_f16, transcribed formula by formula.  It does NOT emulate scheduling, barriers or the DMA queue — it
answers "does every MFMA slot multiply X[b][k] with W[m][k] for the same k, and does every (b, m) see every k exactly once?"
by computing Y through the emulated path and comparing it with a direct product of the same fp16 operands.

    python tools/emul/x4l_layout_check.py          (needs oracle/libggml_oracle.so for valid Q4_K blocks; no GPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R  # noqa: E402

TM, TB, BLK = 256, 128, 144
XS, WQS, TS = TB * 256, TM * 64, TM * 16
ST = XS + WQS + TS
PERM = [0, 2, 1, 3]                                   # the fp16 activation image stores (k0, k2, k1, k3) within every 4


def f16(x):
    return np.float16(x)


def k4_scale_min(scales12, jj):
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


def unpack_pair(src_dword_bytes, pos):
    """nibq(): the two nibbles at bit `pos` of the low and of the high 16 bits of a dword -> (q_lo, q_hi)"""
    lo = int(src_dword_bytes[0]) | (int(src_dword_bytes[1]) << 8); hi = int(src_dword_bytes[2]) | (int(src_dword_bytes[3]) << 8)
    return (lo >> pos) & 15, (hi >> pos) & 15


def chunk_of(kk, h):
    return (kk >> 1) * 4 + 2 * h + (kk & 1)


def main(M=256, B=128, K=512, seed=1):
    rng = np.random.default_rng(seed)
    nsb = K // 256
    w = R.random_weights(R.Q4_K, M, K, seed).reshape(M, nsb, BLK)
    xh = rng.uniform(-1, 1, (B, K)).astype(np.float16)
    # the k-panel-major, pair-interleaved image the kernel reads: [K/128][B][128 halves]
    img = np.zeros((K // 128, B, 128), np.float16)
    for p in range(128):
        img[:, :, p] = xh[:, [pan * 128 + (p & ~3) + PERM[p & 3] for pan in range(K // 128)]].T
    img_b = img.view(np.uint8).reshape(K // 128, B, 256)

    assert M <= TM and B <= TB, "one work-group tile"
    Y = np.zeros((TB, TM), np.float64)                  # the whole tile is computed; rows / columns beyond B / M are not stored
    for st in range(nsb * 2):
        lds = np.zeros(ST, np.uint8)
        # ---- loader waves (mg = loader index, lane): activation pieces
        for li in range(4):
            for lane in range(64):
                for i in range(8):
                    pc = (li + 4 * i) * 64 + lane; row = pc >> 4; c = (pc & 15) ^ (row & 15)
                    lds[(li + 4 * i) * 1024 + lane * 16:][:16] = img_b[st, min(row, B - 1), c * 16:c * 16 + 16]
                for i in range(4):
                    pc = (li + 4 * i) * 64 + lane; row = pc >> 2; c = (pc & 3) ^ ((row >> 2) & 3)
                    src = w[min(row, M - 1), st >> 1, 16 + (st & 1) * 64 + c * 16:][:16]
                    lds[XS + (li + 4 * i) * 1024 + lane * 16:][:16] = src
                lidx = (li << 6) | lane; lrow = lidx >> 1; lgl = lidx & 1
                for r in range(2):
                    row = lrow + 128 * r
                    sl, cl, sh, ch = table_entry(w[min(row, M - 1), st >> 1], (st & 1) * 2 + lgl)
                    lds[XS + WQS + (row * 2 + lgl) * 8:][:8] = np.array([sl, cl, sh, ch], np.float16).view(np.uint8)
        # ---- compute waves
        for mg in range(4):
            for g in range(2):
                for kk in range(4):
                    # fragments of both lane halves
                    for mb in range(2):
                        wf = np.zeros((32, 2, 8), np.float64)            # [j][h][e]
                        for j in range(32):
                            row = mg * 64 + mb * 32 + j
                            te = lds[XS + WQS + (row * 2 + g) * 8:][:8].view(np.float16)
                            s, c = (te[0], te[1]) if kk < 2 else (te[2], te[3])
                            for h in range(2):
                                q = lds[XS + row * 64 + (((2 * g + h) ^ ((row >> 2) & 3)) << 4):][:16]
                                for i in range(4):
                                    dw = (0 if i < 2 else 1) + (2 if (kk & 1) else 0)        # q.x/q.y or q.z/q.w
                                    pos = (4 if kk >= 2 else 0) + (8 if (i & 1) else 0)
                                    qa, qb = unpack_pair(q[dw * 4:dw * 4 + 4], pos)
                                    for e, qv in enumerate((qa, qb)):
                                        wf[j, h, 2 * i + e] = np.float64(f16(np.float64(qv - 8) * np.float64(s) + np.float64(c)))
                        for bf in range(4):
                            xa = np.zeros((32, 2, 8), np.float64)
                            for j in range(32):
                                for h in range(2):
                                    coff = ((g * 8 + chunk_of(kk, h)) ^ (j & 15)) << 4
                                    xa[j, h] = lds[(bf * 32 + j) * 256 + coff:][:16].view(np.float16).astype(np.float64)
                            # v_mfma_f32_32x32x16_f16: D[i][n] += sum_h sum_e A[lane(i,h)][e] * B[lane(n,h)][e]
                            Y[bf * 32:bf * 32 + 32, mg * 64 + mb * 32:mg * 64 + mb * 32 + 32] += np.einsum("ihe,nhe->in", xa, wf)
    # ---- direct product of the same fp16 operands
    wd = np.zeros((M, K), np.float64)
    for m in range(M):
        for sb in range(nsb):
            blk = w[m, sb]
            for G in range(4):
                sl, cl, sh, ch = table_entry(blk, G)
                qs = blk[16 + 32 * G:16 + 32 * G + 32]
                lo = (qs & 15).astype(np.float64) - 8; hi = (qs >> 4).astype(np.float64) - 8
                wd[m, sb * 256 + 64 * G:][:32] = (lo * np.float64(sl) + np.float64(cl)).astype(np.float16)
                wd[m, sb * 256 + 64 * G + 32:][:32] = (hi * np.float64(sh) + np.float64(ch)).astype(np.float16)
    Yd = xh.astype(np.float64) @ wd.T
    err = np.abs(Y[:B, :M] - Yd).max() / np.abs(Yd).max()
    # and the fp16 weights themselves against the reference dequantizer (fp16 rounding only)
    deq = R.o_dequantize(R.Q4_K, w.reshape(-1), K)
    werr = np.abs(wd - deq).max() / np.abs(deq).max()
    print("emulated path vs direct product: max rel err %.3e;  fp16 weights vs dequantize_row_q4_K: %.3e" % (err, werr))
    assert err < 1e-12 and werr < 2e-3
    return err


if __name__ == "__main__":
    main()
