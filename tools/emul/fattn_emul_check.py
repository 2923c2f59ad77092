"""Runs the SOURCE of the FLASH_ATTN_EXT kernel (fattn.hip) on the CPU (tools/emul/fattn_emul) and compares with the CPU oracle
(oracle_flash_attn_ext_f16, the reference's fp16-accumulator semantics) and with a float64 evaluation of the same operator.

    python tools/emul/fattn_emul_check.py [D n_q n_head n_kv]
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
    exe = os.path.join(HERE, "fattn_emul")
    csrc = os.path.join(ROOT, "ggml_amd", "csrc")
    srcs = [os.path.join(HERE, "fattn_emul.cpp"), os.path.join(HERE, "hip_emul.h")] + [os.path.join(csrc, f) for f in ("fattn.hip", "ops.hip", "cdna4_common.h", "cdna4_kernels.h", "epilogue.h", "gemm_q_hw.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-ffp-contract=off", "-I" + os.path.join(HERE, "shim"), "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                        "-Wno-unused-value", "-o", exe, srcs[0]], check=True, capture_output=True, timeout=600)
    return exe


def run(D, n_q, n_head, n_kv, n_head_kv=None, n_batch=1, mask=True, max_bias=0.0, softcap=0.0, permuted=False, inf_every=0, seed=1, timeout=1200, cus=256, mask_unaligned=False, causal=False, ret_y=False):
    """returns (rel-L2 vs the float64 operator, rel-L2 vs the oracle).  cus: the CU count the launcher sizes its grids for (the 128-row
    prefill kernel is chosen when its grid fills the chip: small values select it at test sizes)"""
    n_head_kv = n_head_kv or n_head
    rng = np.random.default_rng(seed)
    q = rng.uniform(-1, 1, (n_batch, n_head, n_q, D)).astype(np.float32)
    k = rng.uniform(-1, 1, (n_batch, n_head_kv, n_kv, D)).astype(np.float16)
    v = rng.uniform(-1, 1, (n_batch, n_head_kv, n_kv, D)).astype(np.float16)
    mrows = (n_q + 63) // 64 * 64
    m = rng.uniform(-1, 1, (mrows, n_kv)).astype(np.float16) if mask else None
    if mask and inf_every:
        m[:, ::inf_every] = -np.inf
        m[0, : n_kv // 2] = -np.inf                    # a query row whose first chunks are masked entirely
    if mask and causal:                                # key j visible to query i iff j <= i + (n_kv - n_q): whole chunks are -inf for whole tiles
        jj, ii = np.meshgrid(np.arange(n_kv), np.arange(mrows))
        m[jj > ii + (n_kv - n_q)] = -np.inf
    scale = 1.0 / np.sqrt(D)
    lay = (lambda a: np.ascontiguousarray(a.transpose(0, 2, 1, 3))) if permuted else (lambda a: a)
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        lay(q).tofile(f("q")); lay(k).tofile(f("k")); lay(v).tofile(f("v"))
        if mask:
            m.tofile(f("m"))
        r = subprocess.run([build()] + [str(x) for x in (D, n_q, n_head, n_batch, n_kv, n_head_kv, n_batch, int(mask), mrows, repr(float(scale)), max_bias, softcap, int(permuted))] +
                           [f("q"), f("k"), f("v"), f("m"), f("o")], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, EMU_CUS=str(cus)))
        if r.returncode == 77:
            return None
        assert r.returncode == 0, r.stderr
        y = np.fromfile(f("o"), np.float32).reshape(n_batch, n_q, n_head, D)
    assert np.isfinite(y).all()
    ye = R.exact_flash_attn_ext(q, k, v, m, scale, max_bias, softcap)
    yo = R.o_flash_attn_ext(q, k, v, m, scale, max_bias, softcap)
    if ret_y:
        return R.rel_l2(y, ye), R.rel_l2(y, yo), y
    return R.rel_l2(y, ye), R.rel_l2(y, yo)


def run_quantized(t, D, n_q, n_head, n_kv, n_head_kv=None, permuted=False, seed=1, timeout=1200, cus=256):
    """K / V as block-quantized rows of ggml type t (the conversion pass k_q_to_f16_dense + the F16 kernels through ggml_cdna4_op_flash_attn_ext):
    returns (rel-L2 vs the float64 operator on the dequantized K / V, bit-identity with the F16 path on fp16(to_float(K)), fp16(to_float(V)))"""
    n_head_kv = n_head_kv or n_head
    rng = np.random.default_rng(seed)
    q = rng.uniform(-1, 1, (1, n_head, n_q, D)).astype(np.float32)
    rows, rb = n_head_kv * n_kv, R.row_size(t, D)
    kb, vb = R.random_weights(t, rows, D, seed=seed + 1), R.random_weights(t, rows, D, seed=seed + 2)
    kf, vf = R.o_dequantize(t, kb, D).reshape(1, n_head_kv, n_kv, D), R.o_dequantize(t, vb, D).reshape(1, n_head_kv, n_kv, D)
    mrows = (n_q + 63) // 64 * 64
    m = rng.uniform(-1, 1, (mrows, n_kv)).astype(np.float16)
    scale = 1.0 / np.sqrt(D)
    lay = (lambda a: np.ascontiguousarray(a.transpose(0, 2, 1, 3))) if permuted else (lambda a: a)
    outs = []
    for kv_type, kk, vv in ((int(t), kb.reshape(1, n_head_kv, n_kv, rb), vb.reshape(1, n_head_kv, n_kv, rb)), (1, kf.astype(np.float16), vf.astype(np.float16))):
        with tempfile.TemporaryDirectory() as d:
            f = lambda n: os.path.join(d, n)
            lay(q).tofile(f("q")); lay(kk).tofile(f("k")); lay(vv).tofile(f("v")); m.tofile(f("m"))
            r = subprocess.run([build()] + [str(x) for x in (D, n_q, n_head, 1, n_kv, n_head_kv, 1, 1, mrows, repr(float(scale)), 0.0, 0.0, int(permuted))] +
                               [f("q"), f("k"), f("v"), f("m"), f("o"), str(kv_type)], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, EMU_CUS=str(cus)))
            if r.returncode == 77:
                return None
            assert r.returncode == 0, r.stderr
            outs.append(np.fromfile(f("o"), np.float32).reshape(1, n_q, n_head, D))
    assert np.isfinite(outs[0]).all()
    return R.rel_l2(outs[0], R.exact_flash_attn_ext(q, kf, vf, m, scale)), bool(np.array_equal(outs[0], outs[1]))


if __name__ == "__main__":
    D, nq, nh, nkv = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 5, 2, 96)
    print("flash-attn source on the CPU, D=%d n_q=%d n_head=%d n_kv=%d: rel-L2 vs float64 / vs oracle =" % (D, nq, nh, nkv), run(D, nq, nh, nkv))
