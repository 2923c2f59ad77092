"""Runs kernel SOURCES on the CPU: tools/emul/<name>_emul.cpp includes a kernel of ggml_amd/csrc and is built with the ROCm
clang as a HOST compiler (hip_emul.h: one OS thread per GPU thread, MFMA / LDS-DMA / counted waits emulated); the output is
compared with a direct product of the same fp16 operands.  What executes is the C++ of the kernel itself — loop bounds, barrier
counts of loader vs compute waves (a mismatch hangs: callers use a timeout), register-array indexing, the epilogue.

    python tools/emul/emul_check.py [M K B [splitk [w12|w8|t64 [exp]]]]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R  # noqa: E402
import layout_ref as LC  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


import importlib.util as _ilu
import sys as _sys
_blm = _sys.modules.get("cdna4_emul_buildlock")                        # (one instance per process: its lock is re-entrant by a process-wide depth count)
if _blm is None:
    _bl = _ilu.spec_from_file_location("cdna4_emul_buildlock", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "buildlock.py"))
    _blm = _ilu.module_from_spec(_bl); _sys.modules["cdna4_emul_buildlock"] = _blm; _bl.loader.exec_module(_blm)
_locked = _blm.locked          # (xdist workers share build/: one build at a time)


@_locked
def build(name="w12"):
    exe = os.path.join(HERE, name + "_emul")
    srcs = [os.path.join(HERE, name + "_emul.cpp"), os.path.join(HERE, "hip_emul.h")] + [os.path.join(ROOT, "ggml_amd", "csrc", f)
            for f in ("gemm_r8.inc", "gemm_kq_t64.inc", "gemm_kq_w12.inc", "gemm_kq_w8.inc", "gemm_w8_epilogue.inc", "gemm_q_hw.h", "gemm_q_common.h", "cdna4_common.h", "cdna4_kernels.h", "quantize_dev.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "ggml_amd", "csrc"),
                        "-Wno-unused-value", "-o", exe, srcs[0]], check=True, capture_output=True, timeout=600)
    return exe


def relayout_image(wtype, w, M, K):
    """numpy restatement of k_repack_q4_0 / k_repack_q8_0 (gemm_q_mfma.hip): the resident 16-byte-aligned images Q4_0R / Q8_0R of M rows of Q4_0 / Q8_0 blocks"""
    nsb = K // 256
    if wtype == R.Q4_0:
        blk = w.reshape(M, nsb, 8, 18)
        out = np.zeros((M, nsb, 144), np.uint8)
        out[:, :, :16] = blk[:, :, :, :2].reshape(M, nsb, 16)
        qs = blk[:, :, :, 2:]                                        # weight j (< 16) = qs[j] & 15, weight 16 + j = qs[j] >> 4
        q = np.concatenate([qs & 15, qs >> 4], axis=3)               # [M][nsb][8][32]
        for g in range(4):
            out[:, :, 16 + 32 * g:48 + 32 * g] = q[:, :, 2 * g] | (q[:, :, 2 * g + 1] << 4)
        return out
    if wtype == R.Q6_K:
        blk = w.reshape(M, nsb, 210)
        out = np.zeros((M, nsb, 288), np.uint8)
        d = blk[:, :, 208:210].copy().view(np.float16).astype(np.float32)                     # [M][nsb][1]
        sc = blk[:, :, 192:208].view(np.int8).astype(np.float32)
        out[:, :, :32] = (d * sc).astype(np.float16).view(np.uint8).reshape(M, nsb, 32)
        ql, qh = blk[:, :, :128], blk[:, :, 128:192]
        for n in range(2):
            for quad in range(4):
                lo = ql[:, :, 64 * n + 32 * (quad & 1):][:, :, :32]
                nib = (lo & 15) if quad < 2 else (lo >> 4)
                q = (nib | (((qh[:, :, 32 * n:32 * n + 32] >> (2 * quad)) & 3) << 4)).astype(np.int16) - 32
                out[:, :, 32 + 128 * n + 32 * quad:][:, :, :32] = q.astype(np.int8).view(np.uint8)
        return out
    blk = w.reshape(M, nsb, 8, 34)
    out = np.zeros((M, nsb, 272), np.uint8)
    out[:, :, :16] = blk[:, :, :, :2].reshape(M, nsb, 16)
    out[:, :, 16:] = blk[:, :, :, 2:].reshape(M, nsb, 256)
    return out


def run_relayout(M, K, B, wtype, seed=1, timeout=900, splitk=1, defer_dma=False, weaken=0):
    """k_gemm_r8<Q4_0R | Q8_0R> on the CPU (lds_emul): Q4_0 / Q8_0 weights through their resident re-layout, against the fp16 weights d * q the kernel builds"""
    rng = np.random.default_rng(seed)
    w = R.random_weights(wtype, M, K, seed)
    xh = rng.uniform(-1, 1, (B, K)).astype(np.float16)
    img = np.zeros((K // 128, B, 128), np.float16)
    for p in range(128):
        img[:, :, p] = xh[:, [pan * 128 + (p & ~3) + LC.PERM[p & 3] for pan in range(K // 128)]].T
    image = relayout_image(wtype, w, M, K)
    if wtype == R.Q6_K:                                             # the kernel multiplies the image's fp16 scale by (q - 32): exact in fp32, one rounding to fp16
        s16 = image[:, :, :32].copy().view(np.float16).astype(np.float32)                      # [M][nsb][16]
        q8 = image[:, :, 32:].view(np.int8).astype(np.float32).reshape(M, K // 256, 16, 16)
        wd = (s16[:, :, :, None] * q8).astype(np.float16).astype(np.float64).reshape(M, K)
    else:
        wd = R.o_dequantize(wtype, w, K).reshape(M, K).astype(np.float16).astype(np.float64)   # d * q: exact in fp32, one rounding to fp16 — as the kernel's packed multiply
    want = xh.astype(np.float64) @ wd.T
    with tempfile.TemporaryDirectory() as d:
        image.tofile(os.path.join(d, "w.bin")); img.tofile(os.path.join(d, "xh.bin"))
        r = subprocess.run([build("lds"), str(M), str(K), str(B), os.path.join(d, "w.bin"), os.path.join(d, "xh.bin"), os.path.join(d, "y.bin"), str(splitk), "256", "3",
                            "102" if wtype == R.Q4_0 else ("108" if wtype == R.Q8_0 else "115")], capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, EMU_DEFER_DMA="1" if defer_dma else "0", EMU_WEAKEN_WAITS=str(weaken)))
        if r.returncode == 77:
            import pytest
            pytest.skip("the environment cannot host the emulation (process / thread limits)")
        assert r.returncode == 0, r.stderr[-500:]
        y = np.fromfile(os.path.join(d, "y.bin"), np.float32).reshape(B, M).astype(np.float64)
    return np.linalg.norm(y - want) / np.linalg.norm(want)


def run(M, K, B, seed=1, timeout=600, splitk=1, kernel="w12", exp=0, xchg_l2=1, return_y=False, defer_dma=False, wtype=None):
    rng = np.random.default_rng(seed)
    nsb = K // 256
    wtype = R.Q4_K if wtype is None else wtype
    q5 = wtype == R.Q5_K
    blk = 176 if q5 else 144
    w = R.random_weights(wtype, M, K, seed).reshape(M, nsb, blk)
    xh = rng.uniform(-1, 1, (B, K)).astype(np.float16)
    img = np.zeros((K // 128, B, 128), np.float16)               # the kernel's activation image: [K/128][B][128], (k0,k2,k1,k3) within every 4
    for p in range(128):
        img[:, :, p] = xh[:, [pan * 128 + (p & ~3) + LC.PERM[p & 3] for pan in range(K // 128)]].T
    wd = np.zeros((M, K), np.float64)                            # the same fp16 weights the kernel builds
    for m in range(M):
        for sb in range(nsb):
            for G in range(4):
                sl, cl, sh, ch = LC.table_entry(w[m, sb], G, 16.0 if q5 else 8.0)       # (header layout is the same: d, dmin, scales[12])
                qs = w[m, sb, (48 if q5 else 16) + 32 * G:][:32]
                lo, hi = (qs & 15).astype(np.float64), (qs >> 4).astype(np.float64)
                if q5:                                                              # fifth bits: qh[l] bit 2G (low nibble) / 2G + 1 (high nibble)
                    qh = w[m, sb, 16:48]
                    lo += 16 * ((qh >> (2 * G)) & 1); hi += 16 * ((qh >> (2 * G + 1)) & 1)
                zero = 16.0 if q5 else 8.0
                wd[m, sb * 256 + 64 * G:][:32] = (lo - zero) * np.float64(sl) + np.float64(cl)
                wd[m, sb * 256 + 64 * G + 32:][:32] = (hi - zero) * np.float64(sh) + np.float64(ch)
    wd = wd.astype(np.float16).astype(np.float64)
    want = xh.astype(np.float64) @ wd.T
    with tempfile.TemporaryDirectory() as d:
        # the weight file must start 16-byte aligned in memory: the emulator reads it into a std::vector (malloc: 16-byte aligned)
        w.tofile(os.path.join(d, "w.bin")); img.tofile(os.path.join(d, "xh.bin"))
        extra = [str(splitk), str(exp), str(xchg_l2), str(wtype)]       # w8: exp = kernel
        r = subprocess.run([build(kernel), str(M), str(K), str(B), os.path.join(d, "w.bin"), os.path.join(d, "xh.bin"), os.path.join(d, "y.bin")] + extra,
                           capture_output=True, text=True, timeout=timeout, env=dict(os.environ, EMU_DEFER_DMA="1" if defer_dma else "0"))
        if r.returncode == 77:                          # process / thread limits of this environment: nothing was checked
            import pytest
            pytest.skip("the environment cannot host the emulation (process / thread limits)")
        assert r.returncode == 0, r.stderr[-500:]
        y = np.fromfile(os.path.join(d, "y.bin"), np.float32).reshape(B, M).astype(np.float64)
    err = np.linalg.norm(y - want) / np.linalg.norm(want)
    return (err, y) if return_y else err



def run_ids(M, K, B, E, seed=1, timeout=900, defer_dma=False):
    """the grouped MUL_MAT_ID form of k_gemm_kq_t64 (IDS = true): E stacked Q4_K experts of M rows, an expert-sorted activation image of
    B rows (tile t -> expert t % E, tile 1 unused, every 7th row padding, output row = B - 1 - r: the tables t64_emul builds under EMU_IDS);
    returns (rel-L2 over the written rows, whether every other output element was left untouched)"""
    rng = np.random.default_rng(seed)
    nsb = K // 256
    w = R.random_weights(R.Q4_K, E * M, K, seed).reshape(E * M, nsb, 144)
    xh = rng.uniform(-1, 1, (B, K)).astype(np.float16)
    img = np.zeros((K // 128, B, 128), np.float16)
    for p in range(128):
        img[:, :, p] = xh[:, [pan * 128 + (p & ~3) + LC.PERM[p & 3] for pan in range(K // 128)]].T
    wd = np.zeros((E * M, K), np.float64)
    for m in range(E * M):
        for sb in range(nsb):
            for G in range(4):
                sl, cl, sh, ch = LC.table_entry(w[m, sb], G, 8.0)
                qs = w[m, sb, 16 + 32 * G:][:32]
                wd[m, sb * 256 + 64 * G:][:32] = ((qs & 15).astype(np.float64) - 8.0) * np.float64(sl) + np.float64(cl)
                wd[m, sb * 256 + 64 * G + 32:][:32] = ((qs >> 4).astype(np.float64) - 8.0) * np.float64(sh) + np.float64(ch)
    wd = wd.astype(np.float16).astype(np.float64)
    want = np.full((B, M), -12345.0)
    for r in range(B):
        t = r // 128
        if t == 1 or r % 7 == 3:
            continue
        e = t % E
        want[B - 1 - r] = xh[r].astype(np.float64) @ wd[e * M:(e + 1) * M].T
    with tempfile.TemporaryDirectory() as d:
        w.tofile(os.path.join(d, "w.bin")); img.tofile(os.path.join(d, "xh.bin"))
        r = subprocess.run([build("t64"), str(M), str(K), str(B), os.path.join(d, "w.bin"), os.path.join(d, "xh.bin"), os.path.join(d, "y.bin"), "1", "128", "0", str(R.Q4_K)],
                           capture_output=True, text=True, timeout=timeout, env=dict(os.environ, EMU_IDS=str(E), EMU_DEFER_DMA="1" if defer_dma else "0"))
        if r.returncode == 77:
            import pytest
            pytest.skip("the environment cannot host the emulation (process / thread limits)")
        assert r.returncode == 0, r.stderr[-500:]
        y = np.fromfile(os.path.join(d, "y.bin"), np.float32).reshape(B, M).astype(np.float64)
    written = want != -12345.0
    return np.linalg.norm(y[written] - want[written]) / np.linalg.norm(want[written]), bool(np.array_equal(y[~written], want[~written]))


if __name__ == "__main__":
    M, K, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 512, 128)
    S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    kern = sys.argv[5] if len(sys.argv) > 5 else "w12"
    exp = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    print("%s kernel source (exp %d) on the CPU vs direct fp16 product, %dx%dx%d split-K %d: rel-L2 %.3e" % (kern, exp, M, K, B, S, run(M, K, B, splitk=S, kernel=kern, exp=exp)))
