"""Runs the SOURCE of the B = 1 decode kernel (k_gemv_q_fused, ggml_amd/csrc/gemv_q.hip) on the CPU (tools/emul/gemv_emul) and
compares with the CPU oracle's MUL_MAT on the same weights and activations.

    python tools/emul/gemv_emul_check.py [type M K]
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


import importlib.util as _ilu
import sys as _sys
_blm = _sys.modules.get("cdna4_emul_buildlock")                        # (one instance per process: its lock is re-entrant by a process-wide depth count)
if _blm is None:
    _bl = _ilu.spec_from_file_location("cdna4_emul_buildlock", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "buildlock.py"))
    _blm = _ilu.module_from_spec(_bl); _sys.modules["cdna4_emul_buildlock"] = _blm; _bl.loader.exec_module(_blm)
_locked = _blm.locked          # (xdist workers share build/: one build at a time)


@_locked
def build():
    exe = os.path.join(HERE, "gemv_emul")
    srcs = [os.path.join(HERE, "gemv_emul.cpp"), os.path.join(HERE, "hip_emul.h")] + [os.path.join(ROOT, "ggml_amd", "csrc", f)
            for f in ("gemv_q.hip", "gemm_q_hw.h", "quantize_dev.h", "cdna4_common.h", "cdna4_kernels.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "ggml_amd", "csrc"),
                        "-Wno-unused-value", "-o", exe, srcs[0]], check=True, capture_output=True, timeout=600)
    return exe


def run(t, M, K, seed=1, timeout=600, env=None, ncol=1):
    """ncol > 1: the one-launch small-batch form (ncol activation rows quantized in LDS by every work-group)"""
    assert K % 1024 == 0, "wave-collective shuffles: the quantizer's lanes must fill whole waves"
    w = R.random_weights(t, M, K, seed)
    x = np.random.default_rng(seed + 1).uniform(-1, 1, (ncol, K)).astype(np.float32)
    want = R.o_mul_mat(t, w, x, M, K)
    want = want[0] if ncol == 1 else want.reshape(-1)
    with tempfile.TemporaryDirectory() as d:
        w.tofile(os.path.join(d, "w.bin")); x.tofile(os.path.join(d, "x.bin"))
        r = subprocess.run([build(), str(t), str(M), str(K), os.path.join(d, "w.bin"), os.path.join(d, "x.bin"), os.path.join(d, "y.bin")] + ([str(ncol)] if ncol > 1 else []),
                           capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
        if r.returncode == 77:
            import pytest
            pytest.skip("the environment cannot host the emulation (process / thread limits)")
        assert r.returncode == 0, r.stderr[-500:]
        y = np.fromfile(os.path.join(d, "y.bin"), np.float32)
    return R.rel_l2(y, want)


def run_cols(t, M, K, ncol, seed=1, timeout=600, staged=False):
    """the multi-column GEMV (B = 2 .. 8) on activations quantized by the oracle, against the oracle's MUL_MAT"""
    w = R.random_weights(t, M, K, seed)
    x = np.random.default_rng(seed + 1).uniform(-1, 1, (ncol, K)).astype(np.float32)
    want = R.o_mul_mat(t, w, x, M, K)
    act = R.o_quantize_act(t, x)
    if R.act_type(t) == R.Q8_K:
        blk = act.reshape(ncol, K // 256, 292)
        d = blk[:, :, 0:4].copy().view(np.float32).reshape(ncol, K // 256)
        qs = blk[:, :, 4:260].copy().view(np.int8).reshape(ncol, K)
        bs = blk[:, :, 260:292].copy().view(np.int16).reshape(ncol, K // 16)
    elif R.act_type(t) == R.Q8_1:                    # {fp16 d, fp16 s, qs[32]}: s travels as fp32 where the K-quants keep their bsums
        blk = act.reshape(ncol, K // 32, 36)
        d = blk[:, :, 0:2].copy().view(np.float16).astype(np.float32).reshape(ncol, K // 32)
        qs = blk[:, :, 4:36].copy().view(np.int8).reshape(ncol, K)
        bs = blk[:, :, 2:4].copy().view(np.float16).astype(np.float32).reshape(ncol, K // 32).view(np.int16)
    else:
        blk = act.reshape(ncol, K // 32, 34)
        d = blk[:, :, 0:2].copy().view(np.float16).astype(np.float32).reshape(ncol, K // 32)
        qs = blk[:, :, 2:34].copy().view(np.int8).reshape(ncol, K)
        bs = np.zeros(0, np.int16)
    with tempfile.TemporaryDirectory() as dd:
        f = lambda n: os.path.join(dd, n)
        w.tofile(f("w.bin")); qs.tofile(f("qs.bin")); d.tofile(f("d.bin")); bs.tofile(f("bs.bin"))
        r = subprocess.run([build(), str(t), str(M), str(K), f("w.bin"), f("qs.bin"), f("y.bin"), str(ncol), f("d.bin"), f("bs.bin")], capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, EMU_STAGED="1") if staged else None)
        if r.returncode == 77:
            import pytest
            pytest.skip("the environment cannot host the emulation (process / thread limits)")
        assert r.returncode == 0, r.stderr[-500:]
        y = np.fromfile(f("y.bin"), np.float32).reshape(ncol, M)
    return R.rel_l2(y, want)


if __name__ == "__main__":
    t, M, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (R.Q4_K, 37, 2048)
    print("decode kernel source on the CPU vs the oracle, type %d %dx%d: rel-L2 %.3e" % (t, M, K, run(t, M, K)))
