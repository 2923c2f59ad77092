"""Torch-free hardware check of the round's widening rows (remaining formats, FLASH_ATTN_EXT) through the C-ABI: device memory straight from
the HIP runtime over ctypes (no 1-2 minute `import torch` on a fresh box), the oracle as the checker.  Writes gpurun_out/widening_check.jsonl.
The pytest versions of the same checks live in tests/test_gpu_widening.py.

    python scripts/gpu_check_widening.py
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import refutil as R  # noqa: E402

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so", mode=C.RTLD_GLOBAL)
L = C.CDLL(os.path.join(ROOT, "ggml_amd", "lib", "libcdna4_kernels.so"))
L.ggml_cdna4_last_error.restype = C.c_char_p
L.ggml_cdna4_mul_mat_workspace_size.restype = C.c_size_t
L.ggml_cdna4_convert_weights_size.restype = C.c_size_t
OUT = os.path.join(ROOT, "gpurun_out", "widening_check.jsonl")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
fails = []


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int32), ("reserved", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4)]


def ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s: %s" % (what, (L.ggml_cdna4_last_error() or b"").decode()))


def dmalloc(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(max(int(n), 256))) == 0
    return p


def to_dev(a):
    a = np.ascontiguousarray(a)
    p = dmalloc(a.nbytes)
    assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
    return p


def to_host(p, shape, dtype):
    a = np.empty(shape, dtype)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, C.c_size_t(a.nbytes), 2) == 0
    return a


def report(**kw):
    with open(OUT, "a") as f:
        f.write(json.dumps(kw) + "\n")
    print(json.dumps(kw), flush=True)
    if not kw.get("ok", True):
        fails.append(kw)


def mul_mat(t, wd, m, k, xd, b, path=0):
    nws = L.ggml_cdna4_mul_mat_workspace_size(C.c_int(t), C.c_int64(k), C.c_int64(b))
    ws, y = dmalloc(nws), dmalloc(4 * m * b)
    ok(L.ggml_cdna4_mul_mat(C.c_int(t), wd, C.c_int64(R.row_size(t, k)), xd, C.c_int64(k), y, C.c_int64(m), C.c_int64(m), C.c_int64(k), C.c_int64(b),
                            ws, C.c_size_t(nws), C.c_int(path), C.c_int(0), C.c_int(0), None), "mul_mat")
    out = to_host(y, (b, m), np.float32)
    hip.hipFree(ws); hip.hipFree(y)
    return out


def check_formats():
    rng = np.random.default_rng(1)
    for name, t, tgt in (("q5_0", R.Q5_0, R.Q8_0), ("q3_K", R.Q3_K, R.Q6_K)):
        for m, k, b in ((16, 256, 9), (130, 768, 33), (512, 2048, 128), (4096, 4096, 512)):
            w = R.random_weights(t, m, k, seed=5 * m + k)
            x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
            wd, xd = to_dev(w), to_dev(x)
            y = mul_mat(t, wd, m, k, xd, b)
            rows = np.arange(m) if m <= 512 else np.random.default_rng(0).choice(m, 64, replace=False)
            rs = R.row_size(t, k)
            wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
            e = R.rel_l2(y[:, rows], R.o_mul_mat(t, wsub, x, len(rows), k))
            # the re-encoding itself, and the target format's GEMM on weights converted up front
            n = L.ggml_cdna4_convert_weights_size(C.c_int(t), C.c_int64(m), C.c_int64(k))
            cd = dmalloc(n)
            ok(L.ggml_cdna4_convert_weights(C.c_int(t), wd, C.c_int64(rs), C.c_int64(m), C.c_int64(k), cd, None), "convert_weights")
            cw = to_host(cd, (n,), np.uint8)
            exact = bool(np.array_equal(R.o_dequantize(tgt, cw, k).view(np.uint32), R.o_dequantize(t, w, k).view(np.uint32))) if m <= 512 else None
            y2 = mul_mat(tgt, cd, m, k, xd, b)
            report(test="more_formats_gemm", type=name, m=m, k=k, b=b, rel_l2=e, reencoding_exact=exact, same_as_target_gemm=bool(np.array_equal(y, y2)),
                   ok=bool(np.isfinite(y).all() and e < 1e-3 and exact is not False and np.array_equal(y, y2)))
            for p in (wd, xd, cd):
                hip.hipFree(p)
    for name, t in (("q5_0", R.Q5_0), ("q2_K", R.Q2_K), ("q3_K", R.Q3_K)):
        m, k, b = 256, 4096, 5
        w = R.random_weights(t, m, k, seed=m + k); x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
        wd, xd = to_dev(w), to_dev(x)
        e = R.rel_l2(mul_mat(t, wd, m, k, xd, b, path=1), R.o_mul_mat(t, w, x, m, k))
        report(test="more_formats_gemv", type=name, m=m, k=k, b=b, rel_l2=e, ok=bool(e < 1e-5))
    for name, t in (("q4_1", R.Q4_1), ("q5_0", R.Q5_0), ("q5_1", R.Q5_1), ("q2_K", R.Q2_K), ("q3_K", R.Q3_K)):
        rows, k = 9, 2048
        w = R.random_weights(t, rows, k, seed=int(t) + 1)
        wd, yd = to_dev(w), dmalloc(4 * rows * k)
        ok(L.ggml_cdna4_dequantize_row(C.c_int(t), wd, yd, C.c_int64(rows * k), None), "dequantize_row")
        got = to_host(yd, (rows, k), np.float32)
        report(test="to_float", type=name, ok=bool(np.array_equal(got.view(np.uint32), R.o_dequantize(t, w, k).view(np.uint32))))


def check_formats2():
    """the formats added after round 2's last hardware session: Q2_K prefill (two-part Q6_K), Q4_1 / Q5_1 (Q8_1 activations; two-part Q8_0 prefill),
    IQ4_NL (Q8_0 re-encoding), IQ4_XS (two-part Q6_K): GEMV units at 1 / 5 rows, the prefill GEMM, to_float, MUL_MAT_ID decode"""
    rng = np.random.default_rng(2)
    for name, t in (("q2_K", R.Q2_K), ("q4_1", R.Q4_1), ("q5_1", R.Q5_1), ("iq4_nl", R.IQ4_NL), ("iq4_xs", R.IQ4_XS), ("q4_0", R.Q4_0), ("q8_0", R.Q8_0)):
        for m, k, b in ((256, 4096, 1), (256, 4096, 5), (16, 256, 9), (130, 768, 33), (512, 2048, 128), (4096, 4096, 512)):
            w = R.random_weights(t, m, k, seed=5 * m + k)
            x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
            wd, xd = to_dev(w), to_dev(x)
            y = mul_mat(t, wd, m, k, xd, b)
            rows = np.arange(m) if m <= 512 else np.random.default_rng(0).choice(m, 64, replace=False)
            rs = R.row_size(t, k)
            wsub = np.concatenate([w[r * rs:(r + 1) * rs] for r in rows])
            e = R.rel_l2(y[:, rows], R.o_mul_mat(t, wsub, x, len(rows), k))
            y2 = mul_mat(t, wd, m, k, xd, b)
            report(test="formats2_mul_mat", type=name, m=m, k=k, b=b, rel_l2=e, deterministic=bool(np.array_equal(y, y2)),
                   ok=bool(np.isfinite(y).all() and e < (1e-5 if b <= 8 else 1e-3) and np.array_equal(y, y2)))
            for p in (wd, xd):
                hip.hipFree(p)
        if R.BLCK[t] == 32:                              # K % 64 == 32: the routing fix of the one-launch decode kernel
            for k in (544, 992):
                w = R.random_weights(t, 48, k, seed=k); x = rng.uniform(-1, 1, (1, k)).astype(np.float32)
                wd, xd = to_dev(w), to_dev(x)
                e = R.rel_l2(mul_mat(t, wd, 48, k, xd, 1), R.o_mul_mat(t, w, x, 48, k))
                report(test="formats2_decode_k_mod_64", type=name, k=k, rel_l2=e, ok=bool(e < 1e-5))
        if t in (R.IQ4_NL, R.IQ4_XS):
            rows, k = 9, 2048
            w = R.random_weights(t, rows, k, seed=int(t) + 1)
            wd, yd = to_dev(w), dmalloc(4 * rows * k)
            ok(L.ggml_cdna4_dequantize_row(C.c_int(t), wd, yd, C.c_int64(rows * k), None), "dequantize_row")
            got = to_host(yd, (rows, k), np.float32)
            report(test="to_float", type=name, ok=bool(np.array_equal(got.view(np.uint32), R.o_dequantize(t, w, k).view(np.uint32))))


def desc(p, type_, es, shape, strides=None):
    """shape / strides in numpy order (slowest first), strides in elements"""
    d = Tensor(); d.data = p.value; d.type = type_; d.reserved = 0
    if strides is None:
        strides = [int(np.prod(shape[i + 1:])) for i in range(4)]
    for i in range(4):
        d.ne[i] = int(shape[3 - i]); d.nb[i] = int(strides[3 - i]) * es
    return d


def check_flash_attn():
    cases = [dict(D=64, n_q=1, n_head=32, n_kv=512), dict(D=128, n_q=35, n_head=32, n_kv=1024), dict(D=256, n_q=32, n_head=32, n_kv=512),
             dict(D=128, n_q=35, n_head=8, n_kv=200, n_head_kv=2, max_bias=8.0), dict(D=128, n_q=3, n_head=4, n_kv=64, softcap=10.0),
             dict(D=256, n_q=33, n_head=4, n_kv=130, n_head_kv=1, mask=False), dict(D=64, n_q=1, n_head=6, n_kv=517, inf_every=7),
             dict(D=128, n_q=40, n_head=4, n_kv=300, n_batch=2, permuted=True), dict(D=128, n_q=512, n_head=8, n_kv=512, n_head_kv=2),
             dict(D=64, n_q=130, n_head=4, n_kv=257, inf_every=4), dict(D=256, n_q=200, n_head=2, n_kv=96, n_batch=2), dict(D=128, n_q=2, n_head=4, n_kv=8192, inf_every=3),
             dict(D=64, n_q=1, n_head=2, n_kv=32768), dict(D=256, n_q=300, n_head=3, n_kv=1000, n_head_kv=1, max_bias=8.0), dict(D=128, n_q=32, n_head=32, n_kv=1024)]
    for c in cases:
        D, n_q, n_head, n_kv = c["D"], c["n_q"], c["n_head"], c["n_kv"]
        n_head_kv, n_batch = c.get("n_head_kv", n_head), c.get("n_batch", 1)
        mask, max_bias, softcap, permuted, inf_every = c.get("mask", True), c.get("max_bias", 0.0), c.get("softcap", 0.0), c.get("permuted", False), c.get("inf_every", 0)
        rng = np.random.default_rng(D + n_q + n_kv)
        q = rng.uniform(-1, 1, (n_batch, n_head, n_q, D)).astype(np.float32)
        k = rng.uniform(-1, 1, (n_batch, n_head_kv, n_kv, D)).astype(np.float16); v = rng.uniform(-1, 1, (n_batch, n_head_kv, n_kv, D)).astype(np.float16)
        mrows = (n_q + 63) // 64 * 64
        m = rng.uniform(-1, 1, (mrows, n_kv)).astype(np.float16) if mask else None
        if mask and inf_every:
            m[:, ::inf_every] = -np.inf; m[0, : n_kv // 2] = -np.inf
        scale = float(1.0 / np.sqrt(D))

        def put(a, type_, es):
            if not permuted:
                return desc(to_dev(a), type_, es, a.shape)
            b_, h_, n_, d_ = a.shape                        # memory order (batch, n, head, D), described as (batch, head, n, D)
            return desc(to_dev(np.ascontiguousarray(a.transpose(0, 2, 1, 3))), type_, es, a.shape, [n_ * h_ * d_, d_, h_ * d_, 1])
        dq, dk, dv = put(q, 0, 4), put(k, 1, 2), put(v, 1, 2)
        dm = desc(to_dev(m), 1, 2, (1, 1, mrows, n_kv)) if mask else None
        od = dmalloc(4 * n_batch * n_q * n_head * D)
        dd = desc(od, 0, 4, (n_batch, n_q, n_head, D))
        t0 = time.time()
        ok(L.ggml_cdna4_op_flash_attn_ext(C.byref(dq), C.byref(dk), C.byref(dv), C.byref(dm) if mask else None, C.byref(dd),
                                          C.c_float(scale), C.c_float(max_bias), C.c_float(softcap), None), "flash_attn_ext")
        y = to_host(od, (n_batch, n_q, n_head, D), np.float32)
        ye, yo = R.exact_flash_attn_ext(q, k, v, m, scale, max_bias, softcap), R.o_flash_attn_ext(q, k, v, m, scale, max_bias, softcap)
        ee, eo, eoe = R.rel_l2(y, ye), R.rel_l2(y, yo), R.rel_l2(yo, ye)
        report(test="flash_attn_ext", **c, rel_l2_float64=ee, rel_l2_oracle=eo, oracle_rel_l2_float64=eoe, first_call_s=round(time.time() - t0, 3),
               ok=bool(np.isfinite(y).all() and ee < 1e-3 and eo < eoe + 1e-3))


def check_flash_attn2():
    """FLASH_ATTN_EXT paths added after round 2's last hardware session: a block-quantized K / V (one conversion pass in front of the F16 kernels) and
    head sizes without a kernel of their own (zero-padded); against a float64 evaluation on the dequantized K / V"""
    cases = [dict(D=128, n_q=1, n_head=8, n_kv=1024, n_head_kv=2, t=R.Q8_0), dict(D=64, n_q=35, n_head=4, n_kv=300, t=R.Q4_0), dict(D=256, n_q=3, n_head=4, n_kv=512, n_head_kv=1, t=R.Q5_1),
             dict(D=128, n_q=200, n_head=8, n_kv=512, t=R.Q4_1), dict(D=80, n_q=35, n_head=8, n_kv=512), dict(D=80, n_q=1, n_head=32, n_kv=1024), dict(D=96, n_q=3, n_head=4, n_kv=200, n_head_kv=2, t=R.Q8_0),
             dict(D=112, n_q=40, n_head=4, n_kv=300), dict(D=80, n_q=512, n_head=8, n_kv=512, n_head_kv=2)]
    for c in cases:
        D, n_q, n_head, n_kv, t = c["D"], c["n_q"], c["n_head"], c["n_kv"], c.get("t")
        n_head_kv = c.get("n_head_kv", n_head)
        rng = np.random.default_rng(D + n_q + n_kv)
        q = rng.uniform(-1, 1, (1, n_head, n_q, D)).astype(np.float32)
        mrows = (n_q + 63) // 64 * 64
        m = rng.uniform(-1, 1, (mrows, n_kv)).astype(np.float16)
        scale = float(1.0 / np.sqrt(D))
        if t is None:
            kf = rng.uniform(-1, 1, (1, n_head_kv, n_kv, D)).astype(np.float16); vf = rng.uniform(-1, 1, (1, n_head_kv, n_kv, D)).astype(np.float16)
            dk, dv = desc(to_dev(kf), 1, 2, kf.shape), desc(to_dev(vf), 1, 2, vf.shape)
        else:
            rows, rb = n_head_kv * n_kv, R.row_size(t, D)
            kb, vb = R.random_weights(t, rows, D, seed=int(t) + D), R.random_weights(t, rows, D, seed=int(t) + D + 1)
            kf, vf = R.o_dequantize(t, kb, D).reshape(1, n_head_kv, n_kv, D), R.o_dequantize(t, vb, D).reshape(1, n_head_kv, n_kv, D)
            dk, dv = desc(to_dev(kb), int(t), R.TYPE_SIZE[t], (1, n_head_kv, n_kv, D // 32)), desc(to_dev(vb), int(t), R.TYPE_SIZE[t], (1, n_head_kv, n_kv, D // 32))
            dk.ne[0] = dv.ne[0] = D                                       # (desc() computed the strides from blocks per row)
        dq, dm = desc(to_dev(q), 0, 4, q.shape), desc(to_dev(m), 1, 2, (1, 1, mrows, n_kv))
        od = dmalloc(4 * n_q * n_head * D)
        dd = desc(od, 0, 4, (1, n_q, n_head, D))
        ok(L.ggml_cdna4_op_flash_attn_ext(C.byref(dq), C.byref(dk), C.byref(dv), C.byref(dm), C.byref(dd), C.c_float(scale), C.c_float(0.0), C.c_float(0.0), None), "flash_attn_ext")
        y = to_host(od, (1, n_q, n_head, D), np.float32)
        ee = R.rel_l2(y, R.exact_flash_attn_ext(q, kf, vf, m, scale))
        report(test="flash_attn_ext2", D=D, n_q=n_q, n_head=n_head, n_kv=n_kv, kv_type=None if t is None else int(t), rel_l2_float64=ee, ok=bool(np.isfinite(y).all() and ee < 1e-3))


def timed(fn, iters=20):
    """average microseconds per call between two HIP events on the null stream"""
    e0, e1 = C.c_void_p(), C.c_void_p()
    hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
    for _ in range(3):
        fn()
    hip.hipEventRecord(e0, None)
    for _ in range(iters):
        fn()
    hip.hipEventRecord(e1, None); hip.hipEventSynchronize(e1)
    ms = C.c_float()
    hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    return ms.value * 1000.0 / iters


def timings(formats=True):
    rng = np.random.default_rng(2)
    m = k = 4096; b = 512
    if formats:
        timings_formats(rng, m, k, b)
    for D, n_q, n_head, n_kv in ((128, 512, 32, 512), (128, 512, 32, 4096), (128, 2048, 32, 2048), (128, 4096, 32, 4096), (128, 1, 32, 4096), (128, 1, 32, 32768), (128, 8, 32, 4096), (64, 512, 32, 1024), (256, 512, 16, 1024)):
        time_fa(rng, D, n_q, n_head, n_kv)
    for name, t in (("q8_0", R.Q8_0), ("q4_0", R.Q4_0)):
        for n_q, n_kv in ((1, 32768), (1, 4096), (8, 4096)):
            time_fa_qkv(rng, 128, n_q, 32, n_kv, name, t)


def timings_formats(rng, m, k, b):
    xd = to_dev(rng.uniform(-1, 1, (b, k)).astype(np.float32))
    for name, t in (("q8_0", R.Q8_0), ("q5_0", R.Q5_0), ("q6_K", R.Q6_K), ("q3_K", R.Q3_K), ("q4_K", R.Q4_K)):
        wd = to_dev(R.random_weights(t, m, k, seed=7))
        nws = L.ggml_cdna4_mul_mat_workspace_size(C.c_int(t), C.c_int64(k), C.c_int64(b))
        ws, y = dmalloc(nws), dmalloc(4 * m * b)
        us = timed(lambda: ok(L.ggml_cdna4_mul_mat(C.c_int(t), wd, C.c_int64(R.row_size(t, k)), xd, C.c_int64(k), y, C.c_int64(m), C.c_int64(m), C.c_int64(k), C.c_int64(b),
                                                   ws, C.c_size_t(nws), C.c_int(0), C.c_int(0), C.c_int(0), None), "mul_mat"))
        report(test="time_mul_mat_step", type=name, m=m, k=k, b=b, us_per_call=round(us, 2), effective_tflops=round(2.0 * m * k * b / us / 1e6, 1))
        for p in (wd, ws, y):
            hip.hipFree(p)


def time_fa_qkv(rng, D, n_q, n_head, n_kv, name, t):
    """decode over a quantized KV cache: the dequantizing operand loads of the key-split kernel against the fp16 copy of round 2 (CDNA4_FA_KV_COPY=1)"""
    q = to_dev(rng.uniform(-1, 1, (1, n_head, n_q, D)).astype(np.float32))
    rb = R.row_size(t, D)
    kk = to_dev(R.random_weights(t, n_head * n_kv, D, seed=3)); vv = to_dev(R.random_weights(t, n_head * n_kv, D, seed=4))
    mrows = (n_q + 63) // 64 * 64
    mm = to_dev(rng.uniform(-1, 1, (mrows, n_kv)).astype(np.float16))
    od = dmalloc(4 * n_q * n_head * D)
    dq, dm, dd = desc(q, 0, 4, (1, n_head, n_q, D)), desc(mm, 1, 2, (1, 1, mrows, n_kv)), desc(od, 0, 4, (1, n_q, n_head, D))
    dk, dv = desc(kk, int(t), rb, (1, n_head, n_kv, 1)), desc(vv, int(t), rb, (1, n_head, n_kv, 1))
    for dsc in (dk, dv):
        dsc.ne[0] = D; dsc.nb[0] = R.type_size(t) if hasattr(R, "type_size") else {int(R.Q8_0): 34, int(R.Q4_0): 18}[int(t)]
    res = {}
    for mode in ("direct", "copy"):
        if mode == "copy":
            os.environ["CDNA4_FA_KV_COPY"] = "1"
        else:
            os.environ.pop("CDNA4_FA_KV_COPY", None)
        res[mode] = timed(lambda: ok(L.ggml_cdna4_op_flash_attn_ext(C.byref(dq), C.byref(dk), C.byref(dv), C.byref(dm), C.byref(dd), C.c_float(0.088), C.c_float(0.0), C.c_float(0.0), None), "fattn"))
    os.environ.pop("CDNA4_FA_KV_COPY", None)
    byts = 2.0 * n_head * n_kv * rb
    report(test="time_flash_attn_ext_quantized_kv", type=name, D=D, n_q=n_q, n_head=n_head, n_kv=n_kv, us_direct=round(res["direct"], 2), us_with_fp16_copy=round(res["copy"], 2),
           cache_GBps_direct=round(byts / res["direct"] / 1e3, 1))
    for p in (q, kk, vv, mm, od):
        hip.hipFree(p)


def time_fa(rng, D, n_q, n_head, n_kv):
    if True:
        q = to_dev(rng.uniform(-1, 1, (1, n_head, n_q, D)).astype(np.float32))
        kk = to_dev(rng.uniform(-1, 1, (1, n_head, n_kv, D)).astype(np.float16)); vv = to_dev(rng.uniform(-1, 1, (1, n_head, n_kv, D)).astype(np.float16))
        mrows = (n_q + 63) // 64 * 64
        mm = to_dev(rng.uniform(-1, 1, (mrows, n_kv)).astype(np.float16))
        od = dmalloc(4 * n_q * n_head * D)
        dq, dk, dv = desc(q, 0, 4, (1, n_head, n_q, D)), desc(kk, 1, 2, (1, n_head, n_kv, D)), desc(vv, 1, 2, (1, n_head, n_kv, D))
        dm, dd = desc(mm, 1, 2, (1, 1, mrows, n_kv)), desc(od, 0, 4, (1, n_q, n_head, D))
        us = timed(lambda: ok(L.ggml_cdna4_op_flash_attn_ext(C.byref(dq), C.byref(dk), C.byref(dv), C.byref(dm), C.byref(dd), C.c_float(0.088), C.c_float(0.0), C.c_float(0.0), None), "fattn"))
        flops = 4.0 * n_head * n_q * n_kv * D
        byts = 2.0 * n_head * n_kv * D * 2 + n_head * n_q * D * 8 + mrows * n_kv * 2
        report(test="time_flash_attn_ext", D=D, n_q=n_q, n_head=n_head, n_kv=n_kv, us_per_call=round(us, 2), tflops=round(flops / us / 1e6, 2), algorithmic_GBps=round(byts / us / 1e3, 1))
        for p in (q, kk, vv, mm, od):
            hip.hipFree(p)


if __name__ == "__main__":
    which = sys.argv[1:] or ["formats", "fattn"]
    for w, fn in (("formats", check_formats), ("formats2", check_formats2), ("fattn", check_flash_attn), ("fattn2", check_flash_attn2), ("timings", timings), ("timings_fa", lambda: timings(False))):
        if w in which:
            try:
                fn()
            except Exception as e:   # noqa: BLE001 — keep going: the other half of the check still says something
                report(test=w, error=repr(e), ok=False)
    print("FAILED: %d" % len(fails) if fails else "ALL OK")
    sys.exit(1 if fails else 0)
