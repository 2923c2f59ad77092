"""The WHOLE kernel library on the CPU (tools/emul/lib_emul.h): ggml_cdna4_mul_mat / _mul_mat_id called through the C-ABI of a host build of
capi.hip + quantize_act.hip + convert_w.hip + gemv_q.hip + gemm_q_mfma.hip + gemm_q_t64.hip + ops.hip — routing, workspace carving, the per-call
re-encodings / re-layouts, the doubled activation image of the two-part formats and every kernel on the route, all from the product's own source —
against the CPU oracle's MUL_MAT.

    python tools/emul/lib_emul_check.py [type M K B [path]]
"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "ggml_amd", "csrc")
OBJ = os.path.join(ROOT, "build", "lib_emul")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refutil as R  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
TUS = ["capi.hip", "quantize_act.hip", "convert_w.hip", "gemv_q.hip", "mmq_i8.hip", "gemm_q_mfma.hip", "gemm_q_t64.hip", "gemm_q_sk.hip", "gemm_q_lds.hip", "ops.hip", "exact.hip"]
FLAGS = ["-std=c++17", "-O1", "-pthread", "-ffp-contract=off", "-I" + os.path.join(HERE, "shim"), "-I" + HERE, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-Wno-unused-value"]


import importlib.util as _ilu
import sys as _sys
_blm = _sys.modules.get("cdna4_emul_buildlock")                        # (one instance per process: its lock is re-entrant by a process-wide depth count)
if _blm is None:
    _bl = _ilu.spec_from_file_location("cdna4_emul_buildlock", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "buildlock.py"))
    _blm = _ilu.module_from_spec(_bl); _sys.modules["cdna4_emul_buildlock"] = _blm; _bl.loader.exec_module(_blm)
_locked = _blm.locked          # (xdist workers share build/: one build at a time)


@_locked
def build():
    os.makedirs(OBJ, exist_ok=True)
    exe = os.path.join(OBJ, "lib_emul")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [os.path.join(HERE, "lib_emul.h"), os.path.join(HERE, "hip_emul.h"), os.path.join(ROOT, "include", "ggml_cdna4.h")]
    newest = max(os.path.getmtime(d) for d in deps)

    def one(f):
        src = os.path.join(CSRC, f) if f.endswith(".hip") else os.path.join(HERE, f)
        obj = os.path.join(OBJ, f + ".o")
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(newest, os.path.getmtime(src)):
            extra = ["-DEMU_DYNAMIC_LDS"] if f == "gemv_q.hip" else []
            inc = ["-x", "c++", "-include", os.path.join(HERE, "lib_emul.h")] if f.endswith(".hip") else []
            subprocess.run([CLANG] + FLAGS + extra + inc + ["-c", src, "-o", obj], check=True, capture_output=True, timeout=900)
            return True
        return False
    with ThreadPoolExecutor(max_workers=8) as ex:
        rebuilt = any(list(ex.map(one, TUS + ["lib_emul_main.cpp"])))
    if rebuilt or not os.path.exists(exe):
        subprocess.run([CLANG, "-pthread", "-o", exe] + [os.path.join(OBJ, f + ".o") for f in TUS + ["lib_emul_main.cpp"]], check=True, capture_output=True, timeout=300)
    return exe


@_locked
def build_so():
    """the same build as a shared library with the library's C-ABI (+ fattn.hip, + cdna4_emul_alloc): tests/emul_torch.py"""
    pic = os.path.join(OBJ, "pic")
    os.makedirs(pic, exist_ok=True)
    out = os.path.join(OBJ, "libcdna4_emul.so")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [os.path.join(HERE, "lib_emul.h"), os.path.join(HERE, "hip_emul.h"), os.path.join(ROOT, "include", "ggml_cdna4.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    tus = TUS + ["fattn.hip", "lib_emul_so.cpp"]

    def one(f):
        src = os.path.join(CSRC, f) if f.endswith(".hip") else os.path.join(HERE, f)
        obj = os.path.join(pic, f + ".o")
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(newest, os.path.getmtime(src)):
            extra = ["-DEMU_DYNAMIC_LDS"] if f == "gemv_q.hip" else []
            inc = ["-x", "c++", "-include", os.path.join(HERE, "lib_emul.h")] if f.endswith(".hip") else []
            subprocess.run([CLANG] + FLAGS + ["-fPIC"] + extra + inc + ["-c", src, "-o", obj], check=True, capture_output=True, timeout=900)
            return True
        return False
    with ThreadPoolExecutor(max_workers=9) as ex:
        rebuilt = any(list(ex.map(one, tus)))
    if rebuilt or not os.path.exists(out):
        subprocess.run([CLANG, "-shared", "-pthread", "-o", out] + [os.path.join(pic, f + ".o") for f in tus], check=True, capture_output=True, timeout=300)
    return out


def _run(args, env, timeout):
    r = subprocess.run(args, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **{k: str(v) for k, v in (env or {}).items()}))
    if r.returncode == 77:
        return False
    assert r.returncode == 0, (r.stdout + r.stderr)[-800:]
    return True


def mul_mat(t, m, k, b, path=0, seed=1, cus=256, timeout=1800, w=None, resident=False, defer_dma=False):
    """-> (rel-L2 of the library's result vs the oracle's MUL_MAT of type t, the result) or None where the environment cannot host the emulation"""
    w = R.random_weights(t, m, k, seed) if w is None else w
    x = np.random.default_rng(seed + 1).uniform(-1, 1, (b, k)).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        w.tofile(f("w")); x.tofile(f("x"))
        if not _run([build(), "mul_mat", str(int(t)), str(m), str(k), str(b), str(path), f("w"), f("x"), f("y")],
                    {"EMU_CUS": cus, "EMU_RESIDENT": int(resident), "EMU_DEFER_DMA": int(defer_dma)}, timeout):
            return None
        y = np.fromfile(f("y"), np.float32).reshape(b, m)
    assert np.isfinite(y).all() and not (y == -12345.0).any(), "unwritten or non-finite outputs"
    return R.rel_l2(y, R.o_mul_mat(t, w, x, m, k)), y


def mul_mat_id(t, m, k, n_expert, n_used, n_b, n_tok, seed=1, cus=256, timeout=1800, resident=False, defer_dma=False, ids=None, env=None, want_y=False):
    """ids: the expert ids [n_tok][n_used] (default: random distinct experts per token); env: extra environment of the run (routing knobs); want_y: -> (rel-L2, y)"""
    rng = np.random.default_rng(seed)
    w = R.random_weights(t, n_expert * m, k, seed)
    xb = rng.uniform(-1, 1, (n_tok, n_b, k)).astype(np.float32)
    if ids is None:
        ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    with tempfile.TemporaryDirectory() as d:
        f = lambda n: os.path.join(d, n)
        w.tofile(f("w")); xb.tofile(f("x")); ids.tofile(f("i"))
        if not _run([build(), "mul_mat_id"] + [str(int(v)) for v in (t, m, k, n_expert, n_used, n_b, n_tok)] + [f("w"), f("x"), f("i"), f("y")],
                    dict({"EMU_CUS": cus, "EMU_RESIDENT": int(resident), "EMU_DEFER_DMA": int(defer_dma)}, **(env or {})), timeout):
            return None
        y = np.fromfile(f("y"), np.float32).reshape(n_tok, n_used, m)
    valid = (ids >= 0) & (ids < n_expert)                                # (a slot with an out-of-range id stays unwritten: the driver's fill value)
    assert np.isfinite(y).all() and not (y[valid] == -12345.0).any() and (y[~valid] == -12345.0).all(), "unwritten, overwritten or non-finite outputs"
    yo = R.o_mul_mat_id(t, w, xb, np.where(valid, ids, 0), m, k, n_expert)
    e = R.rel_l2(y[valid], yo[valid])
    return (e, y) if want_y else e


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    t, m, k, b = a[:4] if len(a) >= 4 else (R.Q4_1, 130, 768, 33)
    r = mul_mat(t, m, k, b, path=a[4] if len(a) > 4 else 0)
    print("library source on the CPU, type %d [%dx%d].[%dx%d]: rel-L2 vs the oracle %s" % (t, m, k, k, b, "n/a (cannot host)" if r is None else "%.3e" % r[0]))
