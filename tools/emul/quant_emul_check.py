"""Runs the SOURCE of the activation quantizers (quantize_act.hip) on the CPU (tools/emul/quant_emul) and compares BIT FOR BIT with the
CPU oracle's quantize_row_q8_K / quantize_row_q8_0 (AVX2 and _ref roundings), and the fp16 activation image of the MFMA GEMM
with fp16(d * q) in its panel-major, pair-interleaved layout.

    python tools/emul/quant_emul_check.py [kind K B]        kind 0 = Q8_K, 1 = Q8_0 (AVX2 rounding), 2 = Q8_0 (_ref rounding), 3 = Q8_1
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
PERM = [0, 2, 1, 3]


import importlib.util as _ilu
import sys as _sys
_blm = _sys.modules.get("cdna4_emul_buildlock")                        # (one instance per process: its lock is re-entrant by a process-wide depth count)
if _blm is None:
    _bl = _ilu.spec_from_file_location("cdna4_emul_buildlock", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "buildlock.py"))
    _blm = _ilu.module_from_spec(_bl); _sys.modules["cdna4_emul_buildlock"] = _blm; _bl.loader.exec_module(_blm)
_locked = _blm.locked          # (xdist workers share build/: one build at a time)


@_locked
def build():
    exe = os.path.join(HERE, "quant_emul")
    srcs = [os.path.join(HERE, "quant_emul.cpp"), os.path.join(HERE, "hip_emul.h")] + [os.path.join(ROOT, "ggml_amd", "csrc", f)
            for f in ("quantize_act.hip", "quantize_dev.h", "cdna4_common.h", "cdna4_kernels.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-ffp-contract=off", "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "ggml_amd", "csrc"),
                        "-Wno-unused-value", "-o", exe, srcs[0]], check=True, capture_output=True, timeout=600)
    return exe


def data(kind_name, shape, seed):
    rng = np.random.default_rng(seed)
    if kind_name == "ties":                       # values on exact .5 grid points: the rounding rules
        return (rng.integers(-254, 255, shape) / 2.0).astype(np.float32)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    x[0, :256] = 0.0                              # an all-zero block
    x[-1, 7] = -np.abs(x[-1, :256]).max() * 2     # a negative maximum
    return x


def run(kind, K, B, dist="uniform", seed=1, timeout=900):
    assert (B * K) % 1024 == 0, "wave-collective shuffles: whole waves only"
    x = data(dist, (B, K), seed)
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        x.tofile(f("x.bin"))
        r = subprocess.run([build(), str(kind), str(K), str(B), f("x.bin"), f("qs.bin"), f("d.bin"), f("bs.bin"), f("xh.bin")], capture_output=True, text=True, timeout=timeout)
        if r.returncode == 77:
            import pytest
            pytest.skip("the environment cannot host the emulation (process / thread limits)")
        assert r.returncode == 0, r.stderr[-500:]
        qs = np.fromfile(f("qs.bin"), np.int8).reshape(B, K)
        qk = 256 if kind == 0 else 32
        dd = np.fromfile(f("d.bin"), np.float32).reshape(B, K // qk)
        bs = np.fromfile(f("bs.bin"), np.int16).reshape(B, K // 16)
        xh = np.fromfile(f("xh.bin"), np.float16).reshape(K // 128, B, 128)
    if kind == 0:
        ref = R.o_quantize_act(R.Q4_K, x).reshape(B, K // 256, 292)
        rd = ref[:, :, 0:4].copy().view(np.float32).reshape(B, K // 256)
        rq = ref[:, :, 4:260].copy().view(np.int8).reshape(B, K)
        rb = ref[:, :, 260:292].copy().view(np.int16).reshape(B, K // 16)
        assert np.array_equal(bs, rb), "bsums"
    elif kind == 3:                                  # Q8_1 (activations of Q4_1 / Q5_1 weights): block {fp16 d, fp16 s, qs[32]}
        ref = R.o_quantize_act(R.Q4_1, x).reshape(B, K // 32, 36)
        rd = ref[:, :, 0:2].copy().view(np.float16).astype(np.float32).reshape(B, K // 32)
        rs = ref[:, :, 2:4].copy().view(np.float16).astype(np.float32).reshape(B, K // 32)
        rq = ref[:, :, 4:36].copy().view(np.int8).reshape(B, K)
        assert np.array_equal(bs.reshape(-1).view(np.float32).reshape(B, K // 32).view(np.uint32), rs.view(np.uint32)), "s = fp16(d * sum q)"
    else:
        ref = R.o_quantize_row("q8_0_cpu" if kind == 1 else "q8_0_ref", x).reshape(B, K // 32, 34)
        rd = ref[:, :, 0:2].copy().view(np.float16).astype(np.float32).reshape(B, K // 32)
        rq = ref[:, :, 2:34].copy().view(np.int8).reshape(B, K)
    assert np.array_equal(qs, rq), "quants"
    assert np.array_equal(dd.view(np.uint32), rd.view(np.uint32)), "scales"
    want = (np.repeat(dd, qk, axis=1) * qs.astype(np.float32)).astype(np.float16)             # fp16(d * q), natural [B][K]
    img = np.zeros_like(xh)
    for p in range(128):
        img[:, :, p] = want[:, [pan * 128 + (p & ~3) + PERM[p & 3] for pan in range(K // 128)]].T
    assert np.array_equal(xh.view(np.uint16), img.view(np.uint16)), "fp16 image"
    return True


if __name__ == "__main__":
    kind, K, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (0, 4096, 8)
    print("quantizer source on the CPU, kind %d, K=%d, B=%d: bit-exact =" % (kind, K, B), run(kind, K, B))
