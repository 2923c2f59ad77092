"""-m gpu: the UNMODIFIED reference harness (tests/test-backend-ops.cpp, compiled by oracle/ref.mk) drives our
plug-in through ggml's own backend ABI and compares every node with the reference CPU backend
(ggml_backend_compare_graph_backend, src/ggml-backend.cpp:1814-1851; NMSE gate 5e-4, :1915-1917)."""
import os
import re
import subprocess
import pytest
import torch
import refutil as R

pytestmark = pytest.mark.gpu
HARNESS = os.path.join(R.REF_DIR, "test-backend-ops")
PLUGIN = os.path.join(R.ROOT, "ggml_amd", "lib", "libggml-cdna4.so")


def _run(op):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    if not (os.path.exists(HARNESS) and os.path.exists(PLUGIN)):
        pytest.fail("prebuilt oracle/_ref/test-backend-ops or ggml_amd/lib/libggml-cdna4.so missing from the snapshot")
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN)
    r = subprocess.run([HARNESS, "test", "-o", op, "-b", "CDNA40"], env=env, capture_output=True, text=True, timeout=900)
    txt = re.sub(r"\x1b\[[0-9;]*m", "", r.stdout + r.stderr)
    os.makedirs(os.path.join(R.ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(R.ROOT, "gpurun_out", "test-backend-ops_%s.log" % op), "w").write(txt)
    return r.returncode, txt


SUPPORTING_OPS = ["ADD", "SUB", "MUL", "DIV", "SCALE", "NORM", "RMS_NORM", "SOFT_MAX", "DIAG_MASK_INF", "GELU", "GELU_QUICK",
                  "SILU", "RELU", "TANH", "GET_ROWS", "CPY", "CONT", "DUP", "ROPE"]


@pytest.mark.parametrize("op", SUPPORTING_OPS)
def test_stock_harness_supporting_ops(op):
    """plain-HIP supporting ops vs ggml-cpu through the reference's own per-op gates (1e-7 NMSE default,
    tests/test-backend-ops.cpp:319-321; CPY 1e-6 :1451-1453; SOFT_MAX 1e-6 :2374-2376)"""
    rc, txt = _run(op)
    n_ok = len(re.findall(r": OK$", txt, re.M))
    assert rc == 0 and "FAIL" not in txt, txt[-4000:]
    assert n_ok >= 1, "no supported case ran for %s\n%s" % (op, txt[-2000:])


@pytest.mark.parametrize("op", ["MUL_MAT", "MUL_MAT_ID"])
def test_stock_harness(op):
    rc, txt = _run(op)
    n_ok = len(re.findall(r": OK$", txt, re.M))
    n_fail = len(re.findall(r"FAIL", txt))
    assert "Backend CDNA40" in txt or "CDNA40" in txt, txt[-3000:]
    assert rc == 0 and n_fail == 0, txt[-4000:]
    assert n_ok > 50, "suspiciously few supported cases ran: %d\n%s" % (n_ok, txt[-3000:])
