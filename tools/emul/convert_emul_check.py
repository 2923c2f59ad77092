"""Runs the SOURCE of the exact weight re-encodings (convert_w.hip: Q5_0 -> Q8_0, Q3_K -> Q6_K, Q2_K -> Q6_K scale part | Q6_K minimum part) on the CPU (tools/emul/convert_emul) and checks
that the oracle's dequantize_row of the RESULT equals its dequantize_row of the SOURCE bit for bit — every 5-bit / 3-bit code, both hmask
polarities, all 64 six-bit scales, any fp16 d (the prefill GEMM of Q5_0 / Q3_K is the Q8_0 / Q6_K GEMM on the re-encoded weights).

    python tools/emul/convert_emul_check.py [type M K]        type 6 = Q5_0, 11 = Q3_K, 10 = Q2_K, 20 = IQ4_NL, 3 / 7 = Q4_1 / Q5_1, 23 = IQ4_XS
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
Q5_0, Q8_0, Q2_K, Q3_K, Q6_K, Q4_1, Q5_1, IQ4_NL, IQ4_XS = 6, 8, 10, 11, 14, 3, 7, 20, 23


import importlib.util as _ilu
import sys as _sys
_blm = _sys.modules.get("cdna4_emul_buildlock")                        # (one instance per process: its lock is re-entrant by a process-wide depth count)
if _blm is None:
    _bl = _ilu.spec_from_file_location("cdna4_emul_buildlock", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "buildlock.py"))
    _blm = _ilu.module_from_spec(_bl); _sys.modules["cdna4_emul_buildlock"] = _blm; _bl.loader.exec_module(_blm)
_locked = _blm.locked          # (xdist workers share build/: one build at a time)


@_locked
def build():
    exe = os.path.join(HERE, "convert_emul")
    srcs = [os.path.join(HERE, "convert_emul.cpp"), os.path.join(HERE, "hip_emul.h")] + [os.path.join(ROOT, "ggml_amd", "csrc", f)
            for f in ("convert_w.hip", "cdna4_common.h", "cdna4_kernels.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-ffp-contract=off", "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "ggml_amd", "csrc"),
                        "-Wno-unused-value", "-o", exe, srcs[0]], check=True, capture_output=True, timeout=600)
    return exe


def source_bytes(t, m, k, seed):
    """fully random block bytes (every code, mask and scale pattern) with finite fp16 scales of both signs, one zero and one subnormal"""
    rng = np.random.default_rng(seed)
    bs, blk, doffs = {Q5_0: (22, 32, [0]), Q3_K: (110, 256, [108]), Q2_K: (84, 256, [80, 82]), Q4_1: (20, 32, [0, 2]), Q5_1: (24, 32, [0, 2]), IQ4_NL: (18, 32, [0]), IQ4_XS: (136, 256, [0])}[t]
    nb = m * k // blk
    raw = rng.integers(0, 256, (nb, bs), dtype=np.uint8)
    for doff in doffs:
        d = rng.uniform(-0.3, 0.3, nb).astype(np.float16)
        d[0] = 0.0
        d[-1] = np.float16(3e-7)
        raw[:, doff:doff + 2] = d.view(np.uint8).reshape(nb, 2)
    return raw.reshape(-1)


def run(t, m, k, seed=1):
    w = source_bytes(t, m, k, seed)
    tgt = Q8_0 if t in (Q5_0, Q4_1, Q5_1, IQ4_NL) else Q6_K
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        w.tofile(f("w.bin"))
        r = subprocess.run([build(), str(t), str(m), str(k), f("w.bin"), f("o.bin")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        out = np.fromfile(f("o.bin"), np.uint8)
    a = R.o_dequantize(t, w, k)
    if t in (Q2_K, Q4_1, Q5_1, IQ4_XS):   # [scale part | minimum part]: 2 K columns per row; dequantize(scale part) + dequantize(minimum part) = dequantize_row of the source
        assert out.size == m * R.row_size(tgt, 2 * k), "size of the re-encoded matrix"
        both = R.o_dequantize(tgt, out, 2 * k)
        b = both[:, :k] + both[:, k:]
        if t in (Q4_1, Q5_1):     # scale part d q; minimum part m in the block's FIRST column only (it meets s = fp16(d_x sum q_x) of the [x~ | s e0] image)
            blkb = w.reshape(m, k // 32, -1)
            mm = blkb[:, :, 2:4].copy().view(np.float16).astype(np.float32).reshape(m, k // 32)
            mpart = both[:, k:].reshape(m, k // 32, 32)
            assert np.array_equal(mpart[:, :, 0].view(np.uint32), mm.view(np.uint32)) and not mpart[:, :, 1:].any(), "minimum part"
            b = both[:, :k] + np.repeat(mm, 32, axis=1)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "dequantize(source) != dequantize(scale part) + m, bit for bit"
            return True
    else:
        assert out.size == m * R.row_size(tgt, k), "size of the re-encoded matrix"
        b = R.o_dequantize(tgt, out, k)
    assert a.shape == b.shape == (m, k)
    if t in (Q2_K, IQ4_XS): # value for value: the sum of the two parts can carry the other sign on a ZERO (x - 0 keeps the sign of x, x + (-0) need not)
        assert np.array_equal(a, b) and np.isfinite(b).all(), "dequantize(source) != dequantize(scale part) + dequantize(minimum part)"
    else:
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "dequantize(source) != dequantize(re-encoded)"
    return True


if __name__ == "__main__":
    t, m, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (Q5_0, 8, 1024)
    print("re-encoding source on the CPU, type %d, M=%d, K=%d: exact =" % (t, m, k), run(t, m, k))
