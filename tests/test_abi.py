"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/ggml_cdna4.h declares, the plug-in exports ggml's DL entry points, host-side geometry helpers agree
with the oracle, and the product fails loudly (no CPU fallback) when asked to compute without a GPU."""
import ctypes as C
import os
import re
import subprocess
import numpy as np
import pytest
import refutil as R

ROOT = R.ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import ggml_amd.native as n
    return n


def test_header_symbols_all_exported(built):
    hdr = open(os.path.join(ROOT, "include", "ggml_cdna4.h")).read()
    declared = set(re.findall(r"\b(ggml_cdna4_[a-z0-9_A-Z]+)\s*\(", hdr))
    declared.discard("ggml_cdna4_type"); declared.discard("ggml_cdna4_path")
    bound = {s[0] for s in built.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    out = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ggml_cdna4_\w+)", out))
    assert declared <= exported, declared - exported
    built.lib()        # and it dlopens + binds


def test_plugin_exports_ggml_entry_points(built):
    if not os.path.exists(built.BACKEND_PATH):
        pytest.skip("plug-in not built (needs the ggml headers)")
    out = subprocess.run(["nm", "-D", "--defined-only", built.BACKEND_PATH], capture_output=True, text=True).stdout
    assert " T ggml_backend_init" in out and " T ggml_backend_score" in out


@pytest.mark.skipif(not os.path.exists(os.path.join(R.REF_DIR, "test-backend-ops")), reason="oracle/_ref not built")
def test_unmodified_reference_harness_loads_plugin(built):
    """the stock test-backend-ops dlopens our .so through GGML_BACKEND_PATH (src/ggml-backend-reg.cpp:577-581)"""
    if not os.path.exists(built.BACKEND_PATH):
        pytest.skip("plug-in not built")
    env = dict(os.environ, GGML_BACKEND_PATH=built.BACKEND_PATH)
    r = subprocess.run([os.path.join(R.REF_DIR, "test-backend-ops"), "test", "-o", "MUL_MAT", "-b", "CDNA40"], env=env, capture_output=True, text=True, timeout=120)
    txt = r.stdout + r.stderr
    assert "failed to find ggml_backend_init" not in txt and "failed to load" not in txt, txt[-2000:]
    assert r.returncode == 0, txt[-2000:]


def test_row_size_matches_oracle(built):
    L = built.lib()
    o = R.oracle()
    o.oracle_row_size.restype = C.c_size_t
    for t in (R.Q4_0, R.Q8_0, R.Q4_K, R.Q5_K, R.Q6_K):
        for k in (256, 4096, 11008 - 11008 % 256):
            assert L.ggml_cdna4_row_size(t, k) == o.oracle_row_size(C.c_int(t), C.c_int64(k)) == R.row_size(t, k)
    assert L.ggml_cdna4_row_size(R.Q4_K, 100) == 0
    from ggml_amd.gtypes import row_size
    assert row_size(12, 4096) == 2304
    with pytest.raises(ValueError):
        row_size(12, 100)


def test_workspace_size_is_monotonic_and_aligned(built):
    L = built.lib()
    prev = 0
    for b in (1, 8, 9, 64, 512):
        n = L.ggml_cdna4_mul_mat_workspace_size(R.Q4_K, 4096, b)
        assert n % 256 == 0 and n >= prev and n >= b * 4096 * 3
        prev = n
    assert L.ggml_cdna4_mul_mat_workspace_size(0, 4096, 8) == 0     # F32 is not a quantized weight type


def test_tail_predictor_is_the_routing_itself_and_needs_no_device(built):
    """ggml_cdna4_mul_mat_fused_residual_may_alias asks the ROUTING (gemm_q_mfma.hip: cdna4_gemm_q_fuses_tail runs cdna4_launch_gemm_q with a probe armed — no scratch, no
    launch) whether the kernel AUTO would take applies the tail in its store; host logic, callable without a GPU.  Pinned here: every route that stores whole finished tiles says
    1 — the GEMV / int8 matrix-core family at small batches, k_gemm_kq_t64 / k_gemm_r8 (Q4_K, Q5_K), the 128 x 128-tile kernels of Q5_K / Q6_K / Q4_0 / Q8_0, and the GEMMs behind the
    exact re-encodings (Q5_0, Q2_K, IQ4_NL, IQ4_XS ...) — and the older per-lane-load kernels (32-weight formats with K not a multiple of 256) say 0: there the call with an aliased
    residual is refused instead of computing 2 (W x) + b."""
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "ggml_amd", "lib", "libcdna4_kernels.so"))
    f = L.ggml_cdna4_mul_mat_fused_residual_may_alias
    f.restype, f.argtypes = C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64]
    for t in (12, 13, 14, 2, 8, 6, 10, 11, 3, 7, 20, 23):
        for (m, k, b) in ((4096, 4096, 1), (4096, 4096, 4), (4096, 4096, 16), (4096, 4096, 96), (4096, 4096, 512), (32768, 8192, 512), (3072, 768, 64)):
            assert f(t, m, k, b) == 1, (t, m, k, b)
    for t in (2, 8, 6):                                                      # K % 256 != 0: k_gemm_q stores the plain product (k_epilogue behind it)
        assert f(t, 4096, 4032, 96) == 0 and f(t, 4096, 4160, 512) == 0, t
        assert f(t, 4096, 4032, 4) == 1                                        # (few rows: the GEMV's store)


def test_route_table_for_a_256_cu_part(built):
    """ggml_cdna4_mul_mat_route: the kernel AUTO takes per (format, shape), answered by the routing code itself without a device (CDNA4_ASSUME_CUS names the part).  Pins what the
    device-time sweeps of round 4 derived (profiles/r04/batch_sweep.txt, gemm_bench.txt, t64_tiles.txt): 1..4 rows one-launch GEMV, 5..48 rows the int8 matrix cores (Q6_K to 32;
    49..64 for small matrices and Q8_0 / Q5_K), above that k_gemm_kq_t64 for Q4_K — k_gemm_r8 once 256 x 256 tiles fill the chip, also for Q5_K —, the 128 x 128-tile kernels for
    the other formats, the older per-lane-load GEMM only for K % 256 != 0; re-encoded formats follow their target (+ 100)."""
    import subprocess, sys, json
    code = r"""
import ctypes as C, json, sys
L = C.CDLL(sys.argv[1])
f = L.ggml_cdna4_mul_mat_route; f.restype = C.c_int; f.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64]
out = {}
for t in (12, 13, 14, 2, 8, 6, 10, 23):
    out[t] = [f(t, 4096, 4096, b) for b in (1, 2, 4, 6, 16, 48, 64, 96, 512)] + [f(t, 32768, 8192, 512), f(t, 4096, 14336, 64), f(t, 4096, 4032, 96), f(t, 16384, 8192, 512), f(t, 4096, 14336, 4)]
print(json.dumps(out))
"""
    def routes(**env):
        r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "ggml_amd", "lib", "libcdna4_kernels.so")], capture_output=True, text=True, timeout=120,
                           env=dict({k: v for k, v in os.environ.items() if k not in ("GGML_CDNA4_OWNED_DEVICE", "GGML_CDNA4_SHARED_DEVICE")}, CDNA4_ASSUME_CUS="256", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", **env))
        assert r.returncode == 0, r.stderr[-800:]
        return {int(k): v for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()}
    got = routes()
    #            rows at 4096^2:  1  2  4  6  16  48  64  96  512 |  C5  4096x14336x64  K=4032x96  16384x8192x512  4096x14336x4
    # (round 6: the DEFAULT never takes a route that waits for a co-resident work-group — the headline shape is quantizer + k_gemm_kq_t64 (10); a host that owns the device
    #  opts in, GGML_CDNA4_OWNED_DEVICE=1 / ggml_cdna4_set_shared_device(0), and gets round 5's ONE launch (11: the quantizer and a grid barrier inside k_gemm_kq_t64<.., FQ>))
    assert got[12] == [1, 1, 1, 3, 3, 3, 3, 10, 10, 12, 10, 0, 10, 3]
    assert routes(GGML_CDNA4_OWNED_DEVICE="1")[12] == [1, 1, 1, 3, 3, 3, 3, 10, 11, 12, 10, 0, 10, 3]
    assert routes(GGML_CDNA4_SHARED_DEVICE="0")[12][8] == 11 and routes(GGML_CDNA4_OWNED_DEVICE="1", GGML_CDNA4_SHARED_DEVICE="1")[12][8] == 10      # (the older knob wins, either way)
    assert got[13] == [1, 1, 1, 3, 3, 3, 3, 13, 13, 12, 3, 0, 13, 3]
    assert got[14] == [1, 1, 1, 3, 3, 13, 13, 13, 13, 13, 13, 0, 13, 3]
    assert got[2] == [1, 1, 1, 3, 3, 3, 3, 13, 13, 13, 13, 14, 13, 3]
    assert got[8] == [1, 1, 1, 3, 3, 3, 3, 13, 13, 13, 3, 14, 13, 3]
    assert got[6] == [1, 2, 2, 2, 103, 103, 103, 113, 113, 113, 103, 114, 113, 2]
    assert got[10][:9] == [1, 2, 2, 2, 113, 113, 113, 113, 113] and got[23][9] == 113


def test_act_image_keys_for_a_256_cu_part(built):
    """ggml_cdna4_act_image_key (round 5: the hand-off of quantized activations between MUL_MATs of one src1): which image a call leaves in its workspace.  Host logic.  Equal
    non-zero keys = the same image: K-quants share the Q8_K fp16 image on the GEMM routes whatever M is (wq / wk / wv of a grouped-query layer), Q4_0 / Q8_0 the Q8_0 one;
    the int8 image of the matrix-core kernel (5..48 rows) is another key; one-launch decode forms leave none (0); a shape where two formats take different routes
    (64 rows: Q8_0 on the int8 matrix cores, Q4_0 on the fp16 GEMM) has different keys; two-part re-encodings (Q2_K) double the image: their own key."""
    import subprocess, sys, json
    code = r"""
import ctypes as C, json, sys
L = C.CDLL(sys.argv[1])
f = L.ggml_cdna4_act_image_key; f.restype = C.c_uint32; f.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64]
print(json.dumps({"kq_gemm": [f(12, 4096, 4096, 512), f(12, 1024, 4096, 512), f(14, 1024, 4096, 512), f(13, 14336, 4096, 512)],
                  "q80_gemm": [f(2, 4096, 4096, 512), f(8, 1024, 4096, 512)], "decode": [f(12, 4096, 4096, 1), f(12, 4096, 4096, 4), f(2, 4096, 4096, 2)],
                  "mmq": [f(12, 4096, 4096, 16), f(14, 1024, 4096, 16), f(8, 4096, 4096, 16)], "b64": [f(8, 4096, 14336, 64), f(2, 4096, 14336, 64)],
                  "q2_K": f(10, 4096, 4096, 512), "q5_0_mmq": f(6, 4096, 4096, 16), "q5_0_gemm": f(6, 4096, 4096, 512), "bad": f(12, 4096, 100, 512),
                  "of": (lambda g: [g(12, 256, 2304, 4096, 4096, 16), g(12, 258, 2304, 4096, 4096, 16), g(12, 256, 2306, 4096, 4096, 16), g(12, 258, 2304, 4096, 4096, 512), g(14, 258, 3360, 4096, 4096, 16), g(14, 257, 3360, 4096, 4096, 16)])(
                      (lambda h: (setattr(h, "restype", C.c_uint32), setattr(h, "argtypes", [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]), h)[2])(L.ggml_cdna4_act_image_key_of))}))
"""
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "ggml_amd", "lib", "libcdna4_kernels.so")], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, CDNA4_ASSUME_CUS="256", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    assert r.returncode == 0, r.stderr[-800:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["kq_gemm"] == [19, 19, 19, 19] and got["q80_gemm"] == [17, 17]
    assert got["decode"] == [0, 0, 0]
    assert got["mmq"] == [3, 3, 1]
    assert got["b64"] == [1, 17]
    assert got["q2_K"] == 27 and got["q5_0_mmq"] == 0 and got["q5_0_gemm"] == 17 and got["bad"] == 0
    # the key of a CONCRETE matrix: a few-row call on rows that are not 16-byte aligned (Q6_K: not 2-byte aligned) leaves no int8 image; the GEMM image does not care
    assert got["of"] == [3, 0, 0, 19, 3, 0]


def test_no_cpu_fallback(built):
    """without a GPU the product must raise, not compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ggml_amd import ops
    from ggml_amd.native import NativeError
    w = torch.zeros(16 * 144, dtype=torch.uint8)
    with pytest.raises(NativeError):
        ops.QTensor(12, 256, 16, w)
    with pytest.raises(NativeError):
        ops.quantize_row_q8_K(torch.zeros(1, 256))


def test_product_never_imports_oracle():
    """only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ggml_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libggml_oracle" not in src and "ggml_oracle.c" not in src, os.path.join(dirpath, f)
                assert not re.search(r"^\s*(import|from)\s+refutil", src, re.M), os.path.join(dirpath, f)
