#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: effective TFLOPS of the Q4_K MUL_MAT hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--variant V --splitk S   (kernel tuning knobs, 0 = auto)]
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): Q4_K [4096x4096]·[4096x512] per GPU — the configuration the metric is quoted on
(BASELINE.json "Q4_K mul_mat 4096² × batch{1,512}"); the batch-1 decode GEMV of the same matrix is reported
beside it under "decode".  One step = one pass of the hot path over one batch with W and the fp32 activations
already resident in HBM: Q8_K activation quantization (exactly the CPU backend's) + the MFMA GEMM.
Multi-GPU: the weight rows (output features) are sharded over ranks as the reference's split buffer does
(src/ggml-cuda/ggml-cuda.cu:729-742); weak scaling — every rank owns 4096 rows (the N-GPU job is the
[4096N x 4096] matrix); the output stays sharded (the consumer of a row-split layer is the next K-split layer),
so the timed data path has no collective; an RCCL all-gather of the output shards is timed separately and
reported as "with_allgather".
The decode step is also timed as 64 nodes of a replayed HIP graph ("decode.hipgraph", computed in a child process so that it
cannot affect the main line; added without a GPU at hand — an "error" field there means the leg failed, nothing else).
The oracle/ reference is used only for the cpu_baseline leg (rank 0, N=1), never inside the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2516.6      # 256 CU x 2.4 GHz x 4096 flop/clk/CU, dense (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0              # spec; ~6300 achievable

M_PER_GPU, K, B = 4096, 4096, 512
Q4_K = 12


def synth_q4k(m, k, seed):
    """random but valid Q4_K superblocks (fp16 d/dmin ~ the scale of uniform(-1,1) weights, random 6-bit
    scales/mins and nibbles) — timing does not depend on the values; parity is tested in tests/."""
    rng = np.random.default_rng(seed)
    nb = m * k // 256
    raw = rng.integers(0, 256, (nb, 144), dtype=np.uint8)
    raw[:, 0:2] = rng.uniform(0.001, 0.004, nb).astype(np.float16).view(np.uint8).reshape(nb, 2)
    raw[:, 2:4] = rng.uniform(0.01, 0.03, nb).astype(np.float16).view(np.uint8).reshape(nb, 2)
    return raw.reshape(-1)


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes of the same command
    (profiles/rNN/pmc_summary.txt: FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE); None if absent."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.txt")))
    if not files:
        return None
    rd = wr = None
    for ln in open(files[-1]):
        if kernel_substr in ln:
            m = re.search(r"read bytes/launch = .*? = ([0-9.]+) MB", ln)
            if m:
                rd = float(m.group(1)) * 1e6
            m = re.search(r"write bytes/launch = .*? = ([0-9.]+) MB", ln)
            if m:
                wr = float(m.group(1)) * 1e6
    return None if rd is None or wr is None else rd + wr


def cpu_baseline(seconds=12.0):
    """the reference CPU backend (oracle/_ref, unmodified ggml built by oracle/ref.mk) on the same workload,
    all host cores; falls back to the C port (oracle/libggml_oracle.so) when the binary is not in the snapshot."""
    cores = os.cpu_count() or 1
    exe = os.path.join(ROOT, "oracle", "_ref", "cpu_baseline")
    if os.path.exists(exe):
        try:
            best = None
            cands = sorted({c for c in (cores, cores // 2, cores // 4, 32, 16) if 1 <= c <= cores}, reverse=True)
            for nt in cands:          # the reference's threadpool is not guaranteed to scale to every SMT thread: report its best
                out = subprocess.run([exe, "q4_K", str(M_PER_GPU), str(K), str(B), str(seconds / len(cands)), str(nt)], capture_output=True, text=True, timeout=seconds * 4 + 60)
                j = json.loads(out.stdout.strip().splitlines()[-1])
                if best is None or j["gflops"] > best["gflops"]:
                    best = j
            return {"value": round(best["gflops"] / 1e3, 4), "unit": "TFLOP/s", "cores": best["threads"], "host_cores": cores, "kind": "reference",
                    "sample": "full workload Q4_K [4096x4096]·[4096x512], ggml-cpu MUL_MAT (unmodified reference, AVX2 build), best of threads=%s: %d runs, %.1f ms/run"
                              % (cands, best["runs"], best["us_per_run"] / 1e3)}
        except Exception as e:  # noqa: BLE001
            print("cpu_baseline(reference) failed: %r" % (e,), file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refutil as R
    w = synth_q4k(512, K, 7)
    x = np.random.default_rng(8).uniform(-1, 1, (B, K)).astype(np.float32)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        R.o_mul_mat(Q4_K, w, x, 512, K); n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(2.0 * 512 * K * B / dt / 1e12, 4), "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": "512 of 4096 weight rows x full [4096x512] activations, oracle C port (OpenMP)"}


# (type id, name, block bytes, weights per block, byte offset of the fp16 scale(s))
FORMATS = [(12, "Q4_K", 144, 256, (0, 2)), (13, "Q5_K", 176, 256, (0, 2)), (14, "Q6_K", 210, 256, (208,)), (2, "Q4_0", 18, 32, (0,)), (8, "Q8_0", 34, 32, (0,))]


def synth_blocks(m, k, seed, bbytes, bweights, scale_offs):
    """random but valid blocks of any of the five formats: random payload bytes, small positive fp16 scales"""
    rng = np.random.default_rng(seed)
    nb = m * k // bweights
    raw = rng.integers(0, 256, (nb, bbytes), dtype=np.uint8)
    for o in scale_offs:
        raw[:, o:o + 2] = rng.uniform(0.001, 0.004, nb).astype(np.float16).view(np.uint8).reshape(nb, 2)
    return raw.reshape(-1)


def format_rows(L, native, ops, dev, stream, x, steps):
    """secondary rows: every weight format at C3' (B=512, GEMM kernel only, activations prepared) and at
    C2 (B=1, the one-launch fused decode step, cache-warm), HIP-event timed"""
    rows = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y = torch.empty((B, M_PER_GPU), dtype=torch.float32, device=dev)
    for tid, name, bb, bw, so in FORMATS:
        a = ops.QTensor.from_host_bytes(tid, K, M_PER_GPU, synth_blocks(M_PER_GPU, K, 99 + tid, bb, bw, so), device=dev)
        ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(tid, K, B), dtype=torch.uint8, device=dev)
        native.check(L.ggml_cdna4_prepare_act(tid, x.data_ptr(), K, K, B, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, stream))

        def gemm():
            native.check(L.ggml_cdna4_mul_mat_prepared(tid, a.data.data_ptr(), a.row_bytes, y.data_ptr(), M_PER_GPU, M_PER_GPU, K, B,
                                                       ws.data_ptr(), ws.numel(), ops.PATH_GEMM, 0, 0, stream))

        def decode():
            native.check(L.ggml_cdna4_mul_mat(tid, a.data.data_ptr(), a.row_bytes, x.data_ptr(), K, y.data_ptr(), M_PER_GPU, M_PER_GPU, K, 1,
                                              ws.data_ptr(), ws.numel(), 0, 0, 0, stream))
        r = {}
        for label, fn in (("gemm_b512_us", gemm), ("decode_b1_us_cache_warm", decode)):
            for _ in range(5):
                fn()
            e0.record()
            for _ in range(steps):
                fn()
            e1.record(); e1.synchronize()
            r[label] = round(e0.elapsed_time(e1) * 1e3 / steps, 3)
        r["gemm_b512_tflops"] = round(2.0 * M_PER_GPU * K * B / (r["gemm_b512_us"] * 1e-6) / 1e12, 1)
        r["weight_bytes"] = a.row_bytes * M_PER_GPU
        rows[name] = r
    return rows


def decode_graph_leg():
    """`bench.py --decode-graph` (a CHILD process of the main run, so that nothing here can take the main JSON line down):
    the B = 1 decode step as it runs inside a captured graph — 64 one-launch GEMVs over 64 rotating matrices (604 MB > the
    Infinity Cache) captured once into a HIP graph and replayed, the way ggml-cuda runs decode
    (src/ggml-cuda/ggml-cuda.cu:2353-2406).  Prints one JSON object."""
    from ggml_amd import native, ops
    L = native.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ncopy = 64
    a = ops.QTensor.from_host_bytes(Q4_K, K, M_PER_GPU, synth_q4k(M_PER_GPU, K, 1234), device=dev)
    big = torch.from_numpy(np.tile(a.data.cpu().numpy().reshape(-1), ncopy)).to(dev)
    row_b, mat_b = a.row_bytes, a.row_bytes * M_PER_GPU
    x1 = torch.from_numpy(np.random.default_rng(4321).uniform(-1, 1, (1, K)).astype(np.float32)).to(dev)
    y1 = torch.empty((1, M_PER_GPU), dtype=torch.float32, device=dev)
    ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(Q4_K, K, 1), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)

    def fused(i):
        native.check(L.ggml_cdna4_mul_mat(Q4_K, big.data_ptr() + (i % ncopy) * mat_b, row_b, x1.data_ptr(), K, y1.data_ptr(), M_PER_GPU, M_PER_GPU, K, 1,
                                          ws.data_ptr(), ws.numel(), 0, 0, 0, torch.cuda.current_stream(dev).cuda_stream))
    with torch.cuda.stream(st):
        for i in range(ncopy):
            fused(i)
    torch.cuda.synchronize(dev)
    y_eager = y1.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for i in range(ncopy):
            fused(i)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(dev)
    same = bool(torch.equal(y1, y_eager))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * ncopy)
    print(json.dumps({"us_per_step_hipgraph": round(us, 3), "nodes_per_graph": ncopy, "same_result_as_stream_launches": same}), flush=True)


T_START = time.time()
OPTIONAL_BUDGET_S = float(os.environ.get("BENCH_OPTIONAL_BUDGET_S", "200"))   # no optional leg STARTS later than this many seconds into the run


def optional_time_left():
    return OPTIONAL_BUDGET_S - (time.time() - T_START)


def diagnostics():
    """Stand-alone probes of tools/microbench (built by __graft_entry__.build()), shape checks and the opt-in tests of code that
    was written after the round's GPU budget was spent — each a child process with its own timeout, run after everything that is
    reported above, most informative first, none started once the optional time budget is spent: what DESIGN.md 7 asks of the
    next GPU call, recorded with the bench line.  Raw text, trimmed."""
    mb = os.path.join(ROOT, "tools", "microbench")
    res = {}

    def tool(name, cmd, env, tmo, slack=0.0):
        if not os.path.exists(os.path.join(mb, cmd[0])):
            res[name] = "not built"
            return
        if optional_time_left() <= -slack:
            res[name] = "skipped: time budget of the optional legs"
            return
        try:
            r = subprocess.run(cmd, cwd=mb, capture_output=True, text=True, timeout=tmo, env=dict(os.environ, **env))
            txt = r.stdout if len(r.stdout) <= 3200 else r.stdout[:1200] + "\n[...]\n" + r.stdout[-2000:]      # (mfma_valu prints its sustained-rate lines first)
            res[name] = (txt if r.returncode == 0 else "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
        except Exception as e:              # noqa: BLE001
            res[name] = repr(e)[:200]

    def shape(name, spec, env):
        if optional_time_left() <= 0:
            res[name] = "skipped: time budget of the optional legs"
            return
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--shape-check", spec], capture_output=True, text=True, timeout=90, env=dict(os.environ, **env))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            res[name] = json.loads(line[-1]) if line else {"returncode": r.returncode, "stderr": (r.stderr or "")[-400:]}
        except Exception as e:              # noqa: BLE001
            res[name] = repr(e)[:200]

    def tests(name, targs):
        if optional_time_left() <= 0:
            res[name] = "skipped: time budget of the optional legs"
            return
        try:
            r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + targs, cwd=ROOT, capture_output=True, text=True, timeout=120,
                               env=dict(os.environ, CDNA4_TEST_EXPERIMENTAL="1"))
            res[name] = {"returncode": r.returncode, "tail": r.stdout[-600:]}
        except Exception as e:              # noqa: BLE001
            res[name] = repr(e)[:200]

    # bit-identical candidates of the shipped GEMM (ablation build): early table read / balanced epilogue / stores from registers / both
    tool("w12_candidates", ["./gemm_bench_abl", "4096", "4096", "512", ""], {"GB_VARIANTS": "4119,69655,135191,266263,397335", "GB_SPLITKS": "0"}, 60)
    # BASELINE configs[2] at its TRUE K = 11008 = 43 superblocks (round 1 measured 10752): today's auto route (no K split for an odd
    # count: 128 of 256 CUs busy) and the uneven 22 / 21 hand-off split behind CDNA4_ODD_SPLIT=1 (emulator-verified), one process each
    shape("c3_4096x11008x512", "4096,11008,512", {"CDNA4_ODD_SPLIT": "0"})
    shape("c3_4096x11008x512_odd_split", "4096,11008,512", {"CDNA4_ODD_SPLIT": "1"})
    # the opt-in parity tests (tests/ run as tests, in a child pytest with CDNA4_TEST_EXPERIMENTAL=1; only the pass / fail summary is
    # kept): in-launch quantizer on small and ragged shapes incl. > 64 launches and the refusal paths; Q5_0 / Q2_K / Q3_K through the
    # GEMV units against the oracle
    tests("experimental_parity_tests", ["tests/test_gpu_parity.py", "-k", "extra_weight_types or fails_loudly or (in_launch_activation and (256-1024 or 300-1536 or 513-3072))"])
    # the default route at the shape of DESIGN 4.3's open issue (own process: a GPU fault there ends only that process)
    shape("default_route_8192x8192x512", "8192,8192,512", {})
    tool("launch_floor", ["./launch_floor"], {}, 60)                    # empty-kernel launch cost; pure-load floor of the 9.4 MB decode matrix
    tool("mfma_valu", ["./mfma_valu", "500"], {}, 60)                   # matrix-pipe price of the unpack mix, the clock held, sustained MFMA-only rate
    tests("experimental_gguf_upload_test", ["tests/test_gguf.py", "-k", "upload"])
    # very last (nothing follows it but the print of the line): the experimental loader-wave kernels of gemm_q_x4l.hip (256x128 / 4 compute
    # waves, 128x128, 256x128 / 8 compute waves; emulator-verified, never run on a GPU) beside the default at the headline shape — time and
    # rel-L2 against the default (must be ~1e-7); one process per form: a fault of one does not hide the others.  (Short: allowed to start
    # a little past the budget.)
    for v in (8199, 24583, 40967):
        tool("x4l_experimental_%d" % v, ["./gemm_bench_abl", "4096", "4096", "512", ""], {"GB_VARIANTS": "4119,%d" % v, "GB_SPLITKS": "0", "GB_ROUNDS": "2"}, 30, slack=30.0)
    return res


def shape_check_leg(spec):
    """`bench.py --shape-check M,K,B` (a CHILD process): the default GEMM route at a shape against the first, slice-per-barrier
    kernel (variant 5) through the C-ABI.  Used for 8192 x 8192 x 512 — the shape at which an ablation harness process once died
    with a GPU fault (open issue) and which no test of round 1 ran on the default route."""
    from ggml_amd import native, ops
    L = native.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    m, k, b = (int(v) for v in spec.split(","))
    a = ops.QTensor.from_host_bytes(Q4_K, k, m, synth_q4k(m, k, 7), device=dev)
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (b, k)).astype(np.float32)).to(dev)
    ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(Q4_K, k, b), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ys = {}
    for variant in (0, 5):
        y = torch.empty((b, m), dtype=torch.float32, device=dev)
        native.check(L.ggml_cdna4_mul_mat(Q4_K, a.data.data_ptr(), a.row_bytes, x.data_ptr(), k, y.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, variant, 0, st))
        torch.cuda.synchronize(dev)
        ys[variant] = y.double()
    y = torch.empty((b, m), dtype=torch.float32, device=dev)
    native.check(L.ggml_cdna4_prepare_act(Q4_K, x.data_ptr(), k, k, b, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, st))

    def gemm():
        native.check(L.ggml_cdna4_mul_mat_prepared(Q4_K, a.data.data_ptr(), a.row_bytes, y.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, 0, 0, st))
    for _ in range(200):
        gemm()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(50):
            gemm()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
    print(json.dumps({"shape": [m, k, b], "finite": bool(torch.isfinite(ys[0]).all()),
                      "rel_l2_default_vs_variant5": float((ys[0] - ys[5]).norm() / ys[5].norm()),
                      "gemm_us_per_launch_default_route": round(best, 3), "tflops": round(2.0 * m * k * b / best / 1e6, 1),
                      "CDNA4_ODD_SPLIT": os.environ.get("CDNA4_ODD_SPLIT", "")}), flush=True)


FUSEQ_PFW_VARIANT = 4119 | (3072 << 16) # + weight pre-touch under the quantizer
FUSEQ_GRP_VARIANT = 4119 | (9216 << 16)    # one counter per (activation tile, K range) group instead of one for the whole grid
FUSEQ_WBL2_VARIANT = 4119 | (5120 << 16)   # image published by plain stores + an agent-scope release fence (L2 write-back): the textbook form, as a reference
FUSEQ_VARIANT = 4119 | (1024 << 16)     # k_gemm_kq_w12<Q4_K> with the Q8_K activation quantizer inside the launch (explicit, experimental)


def fuseq_leg(steps):
    """`bench.py --fuseq-leg` (a CHILD process of the main run: a failure here is recorded in the main JSON line, never
    propagated): the headline step as ONE launch — the GEMM variant that quantizes its own activations behind a one-way grid
    barrier (verified on the CPU emulator; whether its cross-XCD publication holds on the hardware is exactly what this leg
    reports) — beside the default two-launch step, same process, alternating blocks, with a bit-for-bit comparison over
    fresh activations through the same workspace.  Prints one JSON object."""
    from ggml_amd import native, ops
    L = native.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    a = ops.QTensor.from_host_bytes(Q4_K, K, M_PER_GPU, synth_q4k(M_PER_GPU, K, 1234), device=dev)
    ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(Q4_K, K, B), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    y = torch.empty((B, M_PER_GPU), dtype=torch.float32, device=dev)

    def run(x, variant):
        native.check(L.ggml_cdna4_mul_mat(Q4_K, a.data.data_ptr(), a.row_bytes, x.data_ptr(), K, y.data_ptr(), M_PER_GPU,
                                          M_PER_GPU, K, B, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, variant, 0, stream))
    fused = (FUSEQ_VARIANT, FUSEQ_PFW_VARIANT, FUSEQ_WBL2_VARIANT, FUSEQ_GRP_VARIANT)
    same, finite = {v: True for v in fused}, {v: True for v in fused}
    rng = np.random.default_rng(99)
    for it in range(6):                               # fresh activations every time through the SAME workspace: a stale line would show
        x = torch.from_numpy(rng.uniform(-1, 1, (B, K)).astype(np.float32)).to(dev)
        run(x, 4119); torch.cuda.synchronize(dev); y0 = y.clone()
        for v in fused:
            y.fill_(7.0)
            run(x, v); torch.cuda.synchronize(dev)
            finite[v] = finite[v] and bool(torch.isfinite(y).all())
            same[v] = same[v] and bool(torch.equal(y, y0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = {4119: 1e30, FUSEQ_VARIANT: 1e30, FUSEQ_PFW_VARIANT: 1e30, FUSEQ_WBL2_VARIANT: 1e30, FUSEQ_GRP_VARIANT: 1e30}
    for _ in range(100):
        run(x, 4119)
    for _ in range(4):                                # alternating blocks: no variant owns the warm end of the run
        for v in (4119,) + fused:
            for _ in range(10):
                run(x, v)
            e0.record()
            for _ in range(steps):
                run(x, v)
            e1.record(); e1.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) * 1e3 / steps)
    fl = 2.0 * M_PER_GPU * K * B
    print(json.dumps({"what": "step = fp32 X -> Y of Q4_K [4096x4096]·[4096x512]; two launches (k_quantize_q8_K + k_gemm_kq_w12) vs ONE (k_gemm_kq_w12<Q4_K,true,1024>: in-launch quantizer + grid barrier)",
                      "bit_identical_to_default": same[FUSEQ_VARIANT], "finite": finite[FUSEQ_VARIANT],
                      "us_per_step_two_launches": round(best[4119], 3), "us_per_step_one_launch": round(best[FUSEQ_VARIANT], 3),
                      "tflops_two_launches": round(fl / best[4119] / 1e6, 2), "tflops_one_launch": round(fl / best[FUSEQ_VARIANT] / 1e6, 2),
                      "with_weight_pretouch": {"what": "EXP bit 11: the loader lanes touch the prologue's weight bytes while the activations are quantized",
                                               "bit_identical_to_default": same[FUSEQ_PFW_VARIANT], "finite": finite[FUSEQ_PFW_VARIANT],
                                               "us_per_step_one_launch": round(best[FUSEQ_PFW_VARIANT], 3), "tflops_one_launch": round(fl / best[FUSEQ_PFW_VARIANT] / 1e6, 2)},
                      "grouped_counters": {"what": "EXP bit 13: one counter per (activation tile, K range) group — 8 counters of 32 arrivals at this shape instead of one word taking 256",
                                           "bit_identical_to_default": same[FUSEQ_GRP_VARIANT], "finite": finite[FUSEQ_GRP_VARIANT],
                                           "us_per_step_one_launch": round(best[FUSEQ_GRP_VARIANT], 3), "tflops_one_launch": round(fl / best[FUSEQ_GRP_VARIANT] / 1e6, 2)},
                      "published_by_l2_writeback_fence": {"what": "EXP bit 12: plain image stores + agent-scope release fence (buffer_wbl2) instead of write-through stores — reference for the publication",
                                                          "bit_identical_to_default": same[FUSEQ_WBL2_VARIANT], "finite": finite[FUSEQ_WBL2_VARIANT],
                                                          "us_per_step_one_launch": round(best[FUSEQ_WBL2_VARIANT], 3)}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-graph", action="store_true", help="internal: run only the HIP-graph decode leg and print its JSON")
    ap.add_argument("--shape-check", default="", help="internal: M,K,B — default GEMM route against variant 5 at that shape, one JSON object")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the stand-alone probes of tools/microbench at the end of the run")
    ap.add_argument("--fuseq-leg", action="store_true", help="internal: run only the one-launch (in-kernel activation quantizer) leg and print its JSON")
    args = ap.parse_args()
    if args.decode_graph:
        return decode_graph_leg()
    if args.fuseq_leg:
        return fuseq_leg(max(50, min(args.steps, 200)))
    if args.shape_check:
        return shape_check_leg(args.shape_check)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.gpus > 1 or world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from ggml_amd import native, ops
    native.lib()
    L = native.lib()

    # this rank's row shard of the [4096*world x 4096] Q4_K matrix, and the (replicated) activations
    a = ops.QTensor.from_host_bytes(Q4_K, K, M_PER_GPU, synth_q4k(M_PER_GPU, K, 1234 + rank), device=dev)
    x = torch.from_numpy(np.random.default_rng(4321).uniform(-1, 1, (B, K)).astype(np.float32)).to(dev)
    y = torch.empty((B, M_PER_GPU), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(Q4_K, K, B), dtype=torch.uint8, device=dev)

    def step():
        native.check(L.ggml_cdna4_mul_mat(Q4_K, a.data.data_ptr(), a.row_bytes, x.data_ptr(), K, y.data_ptr(), M_PER_GPU,
                                          M_PER_GPU, K, B, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, args.variant, args.splitk, stream))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    flops_step = 2.0 * M_PER_GPU * K * B * world
    value = flops_step / (ms_per_step * 1e-3) / 1e12

    # ---- the dominant kernel alone (activations prepared once), HIP events on the launch stream -------------
    native.check(L.ggml_cdna4_prepare_act(Q4_K, x.data_ptr(), K, K, B, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, stream))

    base_variant = args.variant & 0xFFFF if (args.variant >> 16) in (1024, 3072) else args.variant   # the in-launch quantizer has no prepared-activation form

    def gemm_only():
        native.check(L.ggml_cdna4_mul_mat_prepared(Q4_K, a.data.data_ptr(), a.row_bytes, y.data_ptr(), M_PER_GPU, M_PER_GPU, K, B,
                                                   ws.data_ptr(), ws.numel(), ops.PATH_GEMM, base_variant, args.splitk, stream))
    for _ in range(5):
        gemm_only()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        gemm_only()
    e1.record(); e1.synchronize()
    gemm_us = e0.elapsed_time(e1) * 1e3 / args.steps
    gemm_tflops = 2.0 * M_PER_GPU * K * B / (gemm_us * 1e-6) / 1e12
    # the same launches on all-zero operands (zero weight bytes, zero activation image): identical instruction stream and memory
    # traffic, almost no toggling in the matrix pipe — the gap to the random-data time is what the chip's power management takes
    # (the working hypothesis for the loop's ceiling).  Not a result: a diagnostic beside the roofline fraction.
    zero_us = None
    if rank == 0:
        try:
            wz = torch.zeros_like(a.data)
            wsz = torch.zeros_like(ws)
            def gemm_zero():
                native.check(L.ggml_cdna4_mul_mat_prepared(Q4_K, wz.data_ptr(), a.row_bytes, y.data_ptr(), M_PER_GPU, M_PER_GPU, K, B,
                                                           wsz.data_ptr(), wsz.numel(), ops.PATH_GEMM, base_variant, args.splitk, stream))
            nz = 40                         # few launches: they carry the same kernel name as the measured ones in a rocprofv3 summary of this run
            for _ in range(10):
                gemm_zero()
            e0.record()
            for _ in range(nz):
                gemm_zero()
            e1.record(); e1.synchronize()
            zero_us = round(e0.elapsed_time(e1) * 1e3 / nz, 3)
            del wz, wsz
        except Exception:               # noqa: BLE001 — optional diagnostic
            zero_us = None

    out = None
    if rank == 0:
        out = {
            "metric": "effective TFLOPS (2*M*N*K), Q4_K mul_mat [4096x4096]x[4096x512]", "value": round(value, 3), "unit": "TFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "Q4_K MUL_MAT [4096x4096]·[4096x512] per GPU; step = Q8_K activation quantize + fp16-MFMA GEMM, W and fp32 X resident in HBM",
                       "M_per_gpu": M_PER_GPU, "K": K, "B": B, "parallelism": "row-split x%d, output left sharded" % world,
                       "gemm_variant": args.variant, "splitk": args.splitk},
            "tokens_per_s": round(B * world / (ms_per_step * 1e-3), 1),
            "roofline": {"bound": "mfma", "kernel": "k_gemm_kq_w12<Q4_K> (128x128 tile, 8 compute + 4 loader waves, cross-stage unpack/MFMA pipeline, split-K=2 symmetric exchange)" if args.variant in (0, 23, 2071, 4119) else "gemm variant %d" % args.variant, "achieved": round(gemm_tflops, 3), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / MFMA_F16_PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic("k_gemm_kq_w12<12") if args.variant in (0, 23, 2071, 4119) else None,
                         "traffic_note": "HBM-side bytes/launch (FETCH_SIZE x2 + WRITE_SIZE) from the rocprofv3 PMC passes of this command, profiles/rNN/pmc_summary.txt",
                         "us_per_launch": round(gemm_us, 3), "algorithmic_flops_per_launch": 2.0 * M_PER_GPU * K * B,
                         "us_per_launch_all_zero_operands": zero_us},
        }

    # ---- batch-1 decode GEMV of the same matrix (HBM roofline), rank 0 only ------------------------------------
    if rank == 0:
        ncopy = 64                                      # 64 x 9.4 MB = 604 MB > 256 MB Infinity Cache: every launch streams from HBM
        big = torch.from_numpy(np.tile(a.data.cpu().numpy().reshape(-1), ncopy)).to(dev)
        row_b, mat_b = a.row_bytes, a.row_bytes * M_PER_GPU
        x1 = x[:1].contiguous()
        y1 = torch.empty((1, M_PER_GPU), dtype=torch.float32, device=dev)
        native.check(L.ggml_cdna4_prepare_act(Q4_K, x1.data_ptr(), K, K, 1, ws.data_ptr(), ws.numel(), ops.PATH_GEMV, stream))

        def gemv(i):
            native.check(L.ggml_cdna4_mul_mat_prepared(Q4_K, big.data_ptr() + (i % ncopy) * mat_b, row_b, y1.data_ptr(), M_PER_GPU, M_PER_GPU, K, 1,
                                                       ws.data_ptr(), ws.numel(), ops.PATH_GEMV, 0, 0, stream))
        def fused(i):                                    # B=1 ggml_cdna4_mul_mat: activation quantizer fused into the GEMV launch
            native.check(L.ggml_cdna4_mul_mat(Q4_K, big.data_ptr() + (i % ncopy) * mat_b, row_b, x1.data_ptr(), K, y1.data_ptr(), M_PER_GPU, M_PER_GPU, K, 1,
                                              ws.data_ptr(), ws.numel(), 0, 0, 0, stream))
        res = {}
        for label, fn, rot in (("cold_hbm", gemv, True), ("cache_warm", gemv, False), ("fused_cold_hbm", fused, True), ("fused_cache_warm", fused, False)):
            for i in range(20):
                fn(i if rot else 0)
            n = max(args.steps, 256)
            e0.record()
            for i in range(n):
                fn(i if rot else 0)
            e1.record(); e1.synchronize()
            res[label] = e0.elapsed_time(e1) * 1e3 / n
        alg_bytes = mat_b + K * 1 + (K // 256) * 4 + (K // 16) * 2 + M_PER_GPU * 4    # W + int8 x + scales + bsums + y
        # full decode step incl. the activation quantize (what graph_compute does for one MUL_MAT node)
        for _ in range(10):
            native.check(L.ggml_cdna4_mul_mat(Q4_K, a.data.data_ptr(), row_b, x1.data_ptr(), K, y1.data_ptr(), M_PER_GPU, M_PER_GPU, K, 1, ws.data_ptr(), ws.numel(), 0, 0, 0, stream))
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(500):
            native.check(L.ggml_cdna4_mul_mat(Q4_K, a.data.data_ptr(), row_b, x1.data_ptr(), K, y1.data_ptr(), M_PER_GPU, M_PER_GPU, K, 1, ws.data_ptr(), ws.numel(), 0, 0, 0, stream))
        torch.cuda.synchronize(dev); full_us = (time.perf_counter() - t0) / 500 * 1e6
        fused_bytes = mat_b + K * 4 + M_PER_GPU * 4                              # W + fp32 x + y (the quantized row never leaves LDS)
        out["decode"] = {"workload": "Q4_K [4096x4096]·[4096x1] (BASELINE configs[1])",
                         "us_per_step": round(res["fused_cold_hbm"], 3), "us_per_step_cache_warm": round(res["fused_cache_warm"], 3),
                         "us_per_step_host_wall": round(full_us, 3), "tokens_per_s": round(1e6 / full_us, 1),
                         "effective_tflops": round(2.0 * M_PER_GPU * K / (res["fused_cold_hbm"] * 1e-6) / 1e12, 3),
                         "note": "step = ggml_cdna4_mul_mat at B=1 = ONE launch (activation quantizer fused into the GEMV); cold = 64 rotating copies of W (604 MB > MALL), HIP-event timed; host_wall includes the Python/ctypes call overhead",
                         "roofline": {"bound": "hbm", "kernel": "k_gemv_q_fused<Q4_K>", "achieved": round(fused_bytes / (res["fused_cold_hbm"] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round(fused_bytes / (res["fused_cold_hbm"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("k_gemv_q_fused<12"),
                                      "algorithmic_bytes_per_launch": fused_bytes},
                         "two_kernel_path": {"us_per_gemv_cold_hbm": round(res["cold_hbm"], 3), "us_per_gemv_cache_warm": round(res["cache_warm"], 3),
                                             "roofline": {"bound": "hbm", "kernel": "k_gemv_q<Q4_K,1> (pre-quantized activations, B=2..8 and MUL_MAT_ID)", "achieved": round(alg_bytes / (res["cold_hbm"] * 1e-6) / 1e9, 1),
                                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / (res["cold_hbm"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                          "traffic": pmc_traffic("k_gemv_q<12, 1"), "algorithmic_bytes_per_launch": alg_bytes}}}
        del big
        # the same decode step replayed from a captured HIP graph (how a decode loop launches it), in a child process: a failure
        # there is recorded, never propagated
        if world == 1:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--decode-graph"], capture_output=True, text=True, timeout=240)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                hg = json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-300:]}
            except Exception as e:          # noqa: BLE001 — any failure of the optional leg is data, not an error of the run
                hg = {"error": repr(e)[:300]}
            if "us_per_step_hipgraph" in hg:
                gb = fused_bytes / (hg["us_per_step_hipgraph"] * 1e-6) / 1e9
                hg["roofline"] = {"bound": "hbm", "kernel": "k_gemv_q_fused<Q4_K>, launched from a HIP graph", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(gb / HBM_PEAK_GBS, 4)}
            out["decode"]["hipgraph"] = hg
        out["formats"] = format_rows(L, native, ops, dev, stream, x, max(50, min(args.steps, 200)))
        # the vendor library's dense fp16 GEMM at the same shape on ALREADY-dequantized weights (torch.matmul -> hipBLASLt/rocBLAS):
        # not part of the product and not a baseline of the metric — a practical ceiling next to the nominal MFMA roof, i.e. what
        # a plain fp16 GEMM of this small shape reaches on this box without any dequantization. A failure is recorded, not raised.
        if world == 1:
            try:
                wh = torch.empty((M_PER_GPU, K), dtype=torch.float16, device=dev).uniform_(-1, 1)
                xh = x.to(torch.float16)
                yh = torch.empty((B, M_PER_GPU), dtype=torch.float16, device=dev)
                for _ in range(20):
                    torch.matmul(xh, wh.t(), out=yh)
                e0.record()
                for _ in range(args.steps):
                    torch.matmul(xh, wh.t(), out=yh)
                e1.record(); e1.synchronize()
                lib_us = e0.elapsed_time(e1) * 1e3 / args.steps
                lib_tf = 2.0 * M_PER_GPU * K * B / (lib_us * 1e-6) / 1e12
                out["library_fp16_gemm"] = {"what": "torch.matmul fp16 [512x4096]·[4096x4096]^T (hipBLASLt), weights pre-dequantized: 33.5 MB of fp16 W instead of 9.4 MB of Q4_K",
                                            "us_per_launch": round(lib_us, 3), "tflops": round(lib_tf, 3), "frac_of_peak": round(lib_tf / MFMA_F16_PEAK_TFLOPS, 4),
                                            "ours_over_library": round(gemm_tflops / lib_tf, 3)}
                del wh, xh, yh
            except Exception as e:          # noqa: BLE001
                out["library_fp16_gemm"] = {"error": repr(e)[:300]}

    # ---- the exchange step of a row-split layer, timed separately: all-gather of the output shards ------------
    if dist is not None:
        yfull = torch.empty((world * B, M_PER_GPU), dtype=torch.float32, device=dev)
        for _ in range(3):
            step(); dist.all_gather_into_tensor(yfull, y)
        barrier(); t0 = time.perf_counter()
        for _ in range(args.steps):
            step(); dist.all_gather_into_tensor(yfull, y)
        barrier(); el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            ms = float(t.item()) / args.steps * 1e3
            out["with_allgather"] = {"ms_per_step": round(ms, 5), "value": round(flops_step / (ms * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                                     "collective": "RCCL all_gather_into_tensor of fp32 output shards, %d B/rank" % (B * M_PER_GPU * 4)}
        # the K-split variant of the same layer: every rank holds W[:, K-shard] (whole superblocks per row),
        # computes a full-size partial Y over its shard — the same [4096x4096]·[4096x512] kernel per rank — and the
        # partials are summed with one RCCL all-reduce
        for _ in range(3):
            step(); dist.all_reduce(y)
        barrier(); t0 = time.perf_counter()
        for _ in range(args.steps):
            step(); dist.all_reduce(y)
        barrier(); el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            ms = float(t.item()) / args.steps * 1e3
            out["with_allreduce"] = {"ms_per_step": round(ms, 5), "value": round(flops_step / (ms * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                                     "collective": "K-split: RCCL all_reduce(sum) of the fp32 partial outputs, %d B" % (B * M_PER_GPU * 4)}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and optional_time_left() <= 0:
            out["one_launch_step_experimental"] = "skipped: time budget of the optional legs"
        elif world == 1:
            # experimental one-launch step (not the default path, not part of `value`), in a child process so that nothing it does
            # can take this line down; runs last
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--fuseq-leg", "--steps", str(args.steps)], capture_output=True, text=True, timeout=150)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                out["one_launch_step_experimental"] = json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-300:], "returncode": r.returncode}
            except Exception as e:          # noqa: BLE001
                out["one_launch_step_experimental"] = {"error": repr(e)[:300]}
            if not args.no_diagnostics:
                out["diagnostics"] = diagnostics()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
