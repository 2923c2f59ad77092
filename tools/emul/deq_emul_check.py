"""Runs the SOURCE of the to_float kernels (ops.hip: deq_elem behind ggml_cdna4_dequantize_row / GET_ROWS / CPY -> F32) on the CPU
(tools/emul/deq_emul) and compares BIT FOR BIT with the CPU oracle's dequantize_row_* on fully random block bytes.

    python tools/emul/deq_emul_check.py [type K]
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
# type id -> byte offsets of the fp16 fields of a block (everything else may be any bit pattern)
F16_FIELDS = {R.Q4_0: [0], R.Q4_1: [0, 2], R.Q5_0: [0], R.Q5_1: [0, 2], R.Q8_0: [0], R.Q2_K: [80, 82], R.Q3_K: [108], R.Q4_K: [0, 2], R.Q5_K: [0, 2], R.Q6_K: [208], R.IQ4_NL: [0], R.IQ4_XS: [0]}


import importlib.util as _ilu
import sys as _sys
_blm = _sys.modules.get("cdna4_emul_buildlock")                        # (one instance per process: its lock is re-entrant by a process-wide depth count)
if _blm is None:
    _bl = _ilu.spec_from_file_location("cdna4_emul_buildlock", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "buildlock.py"))
    _blm = _ilu.module_from_spec(_bl); _sys.modules["cdna4_emul_buildlock"] = _blm; _bl.loader.exec_module(_blm)
_locked = _blm.locked          # (xdist workers share build/: one build at a time)


@_locked
def build():
    exe = os.path.join(HERE, "deq_emul")
    csrc = os.path.join(ROOT, "ggml_amd", "csrc")
    srcs = [os.path.join(HERE, "deq_emul.cpp"), os.path.join(HERE, "hip_emul.h")] + [os.path.join(csrc, f) for f in ("ops.hip", "cdna4_common.h", "cdna4_kernels.h", "epilogue.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-ffp-contract=off", "-I" + os.path.join(HERE, "shim"), "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                        "-Wno-unused-value", "-o", exe, srcs[0]], check=True, capture_output=True, timeout=600)
    return exe


def run(t, k, seed=1):
    rng = np.random.default_rng(seed)
    nb = k // R.BLCK[t]
    raw = rng.integers(0, 256, (nb, R.TYPE_SIZE[t]), dtype=np.uint8)
    for o in F16_FIELDS[t]:
        d = rng.uniform(-0.3, 0.3, nb).astype(np.float16)
        d[0] = 0.0
        raw[:, o:o + 2] = d.view(np.uint8).reshape(nb, 2)
    w = raw.reshape(-1)
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        w.tofile(f("w.bin"))
        r = subprocess.run([build(), str(int(t)), str(k), f("w.bin"), f("y.bin")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        y = np.fromfile(f("y.bin"), np.float32)
    want = R.o_dequantize(t, w, k)[0]
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32)), "to_float differs from the oracle"
    return True


def run_f16(t, k, nrows=6, gap=10, seed=1):
    """k_q_to_f16_dense: rows with a gap between them (a strided quantized K / V) -> dense fp16(to_float), bit for bit"""
    rng = np.random.default_rng(seed)
    nb = k // R.BLCK[t]
    raw = rng.integers(0, 256, (nrows, nb, R.TYPE_SIZE[t]), dtype=np.uint8)
    for o in F16_FIELDS[t]:
        d = rng.uniform(-0.3, 0.3, (nrows, nb)).astype(np.float16)
        d[0, 0] = 0.0
        raw[:, :, o:o + 2] = d.view(np.uint8).reshape(nrows, nb, 2)
    rows = raw.reshape(nrows, -1)
    padded = np.concatenate([rows, np.full((nrows, gap), 0x5A, np.uint8)], axis=1)
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        padded.tofile(f("w.bin"))
        r = subprocess.run([build(), "f16", str(int(t)), str(k), str(nrows), str(gap), f("w.bin"), f("y.bin")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        y = np.fromfile(f("y.bin"), np.float16).reshape(nrows, k)
    want = R.o_dequantize(t, rows.reshape(-1), k).astype(np.float16)
    assert np.array_equal(y.view(np.uint16), want.view(np.uint16)), "fp16 copy differs from fp16(oracle to_float)"
    return True


def run_cpyq(t, k=256, nrows=5, dist="uniform", seed=1):
    """the CPY quantizers (k_cpy_f32_to_q / k_cpy_f32_to_q45: F32 -> Q4_0 / Q8_0 / Q4_1 / Q5_0 / Q5_1) against the oracle's quantize_row_*_ref, byte for byte"""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (nrows, k)).astype(np.float32) if dist == "uniform" else (rng.standard_normal((nrows, k)) * 3).astype(np.float32)
    x[0, :32] = 0.0                                   # an all-zero block
    x[1, 32:64] = 0.75                                # a constant block (max == min)
    x[2, 5] = -np.abs(x[2, :32]).max() * 2            # a negative maximum
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        x.tofile(f("x.bin"))
        r = subprocess.run([build(), "cpyq", str(int(t)), str(k), str(nrows), f("x.bin"), f("y.bin")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        y = np.fromfile(f("y.bin"), np.uint8)
    name = {R.Q4_0: "q4_0_ref", R.Q8_0: "q8_0_ref", R.Q4_1: "q4_1_ref", R.Q5_0: "q5_0_ref", R.Q5_1: "q5_1_ref"}[t]
    want = np.concatenate([R.o_quantize_row(name, x[i]) for i in range(nrows)])
    assert np.array_equal(y, want), "CPY quantizer differs from the oracle's %s" % name
    return True


if __name__ == "__main__":
    t, k = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (R.Q3_K, 2048)
    print("to_float source on the CPU, type %d, K=%d: bit-exact =" % (t, k), run(t, k))
