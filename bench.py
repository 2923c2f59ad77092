#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: effective TFLOPS (2*M*N*K) + tokens/s of the Q4_K MUL_MAT hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|c5] [--variant V --splitk S (kernel knobs, 0 = auto)]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; one rank per GPU, RCCL.
   Started WITHOUT a launcher — `python bench.py --gpus N`, WORLD_SIZE unset — it re-executes itself under torch.distributed.run with N ranks: round 6.)

config headline (default; the shape the metric is quoted on, BASELINE configs[1] / north-star target):
    per GPU Q4_K [4096x4096]·[4096x512]; one step = one pass of the hot path over one batch with W and the fp32 activations
    resident in HBM = Q8_K activation quantization (exactly the CPU backend's) + the fp16-MFMA GEMM, through the C-ABI.
    N GPUs: the N-GPU job is the [4096 N x 4096] matrix row-split as the reference's split buffer does
    (src/ggml-cuda/ggml-cuda.cu:729-742) — weak scaling, output left sharded, no collective in the timed path.
config c5 (BASELINE configs[4]): Q4_K [32768x8192]·[8192x512] STRONG scaling: rank r owns rows [32768 r / N, 32768 (r+1) / N);
    value = 2*32768*8192*512 / (max-over-ranks time); also reported with the RCCL all-gather of the output (fp32 and fp16) and
    as the K-split variant (rank r owns whole superblocks [r K/N, (r+1) K/N) of every row, RCCL all-reduce of the partial
    outputs) whose result is CHECKED against the row-split result.  A default multi-GPU run also carries the c5 numbers ("c5").
Inputs: W = fp32 uniform(-1,1) from std::mt19937(1234) through the reference's ggml_quantize_chunk, X =
    uniform(-1,1) from mt19937(4321) — produced by oracle/_ref/synth_data (measurement infrastructure, never in the timed
    region); when that binary is absent the weights are random VALID blocks and "data" says so.
Beside the metric: roofline of the dominant kernel (HIP events on the launch stream) with the vendor library's fp16 GEMM of the
same shape as "library_ceiling", the C3 / C5 shapes, the decode GEMV (4096x4096 and the reference's perf shape 4096x14336),
all weight formats, the reference CPU backend on the host cores (thread sweep, AVX2 and AVX-512 builds), the stock
test-backend-ops perf lines for the plug-in.  oracle/ is touched only by the cpu_baseline / synth legs.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2516.6      # 256 CU x 2.4 GHz x 4096 flop/clk/CU, dense (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0              # spec; ~6300 achievable
Q4_K = 12
HEAD = (4096, 4096, 512)           # M per GPU, K, B
C5 = (32768, 8192, 512)
REFDIR = os.path.join(ROOT, "oracle", "_ref")
T_START = time.time()


def time_left(budget):
    return budget - (time.time() - T_START)


# ------------------------------------------------------------------------------------------------ synthetic inputs
def synth_blocks(t, m, k, seed):
    """fallback only (no oracle/_ref/synth_data): random but VALID blocks with small positive fp16 scales"""
    geo = {12: (144, 256, (0, 2)), 13: (176, 256, (0, 2)), 14: (210, 256, (208,)), 2: (18, 32, (0,)), 8: (34, 32, (0,)),
           6: (22, 32, (0,)), 10: (84, 256, (80, 82)), 11: (110, 256, (108,)), 3: (20, 32, (0, 2)), 7: (24, 32, (0, 2)), 20: (18, 32, (0,)), 23: (136, 256, (0,))}[t]
    rng = np.random.default_rng(seed)
    nb = m * k // geo[1]
    raw = rng.integers(0, 256, (nb, geo[0]), dtype=np.uint8)
    for o in geo[2]:
        raw[:, o:o + 2] = rng.uniform(0.001, 0.004, nb).astype(np.float16).view(np.uint8).reshape(nb, 2)
    return raw.reshape(-1)


TYPE_NAME = {12: "q4_K", 13: "q5_K", 14: "q6_K", 2: "q4_0", 8: "q8_0", 6: "q5_0", 10: "q2_K", 11: "q3_K", 3: "q4_1", 7: "q5_1", 20: "iq4_nl", 23: "iq4_xs"}


def prescribed(t, m, k, row_lo, row_hi, b):
    """(W bytes of rows [row_lo, row_hi) of the mt19937(1234) matrix quantized by the reference, X fp32 [b][k], how) — see module doc"""
    exe = os.path.join(REFDIR, "synth_data")
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "s")
            r = subprocess.run([exe, TYPE_NAME[t], str(m), str(k), str(row_lo), str(row_hi), str(b), p], capture_output=True, text=True, timeout=600)
            if r.returncode == 0:
                return np.fromfile(p + ".w.bin", np.uint8), np.fromfile(p + ".x.bin", np.float32).reshape(b, k), "prescribed"
            print("synth_data failed: %s" % r.stderr[-300:], file=sys.stderr)
    x = np.random.default_rng(4321).uniform(-1, 1, (b, k)).astype(np.float32)
    return synth_blocks(t, row_hi - row_lo, k, 1234 + row_lo), x, "random-valid-blocks"


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of a kernel from the latest committed PMC passes of this command (profiles/rNN/pmc_summary.txt:
    FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE); None if absent."""
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


# ------------------------------------------------------------------------------------------------ the C-ABI, thinly wrapped
class Hot:
    """one weight shard + activations resident in HBM; step() = the hot path through the C-ABI (ggml_cdna4_mul_mat)"""

    def __init__(self, dev, t, w_bytes, m, k, x, variant=0, splitk=0):
        from ggml_amd import native, ops
        self.L, self.native, self.ops = native.lib(), native, ops
        self.dev, self.t, self.m, self.k, self.b = dev, t, m, k, x.shape[0]
        self.a = ops.QTensor.from_host_bytes(t, k, m, w_bytes, device=dev)
        self.x = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        self.y = torch.empty((self.b, m), dtype=torch.float32, device=dev)
        self.ws = torch.empty(self.L.ggml_cdna4_mul_mat_workspace_size(t, k, self.b), dtype=torch.uint8, device=dev)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.variant, self.splitk = variant, splitk
        self.flops = 2.0 * m * k * self.b

    def step(self):
        self.native.check(self.L.ggml_cdna4_mul_mat(self.t, self.a.data.data_ptr(), self.a.row_bytes, self.x.data_ptr(), self.k, self.y.data_ptr(), self.m,
                                                    self.m, self.k, self.b, self.ws.data_ptr(), self.ws.numel(), self.ops.PATH_GEMM if self.b > 8 else 0, self.variant, self.splitk, self.stream))

    def prepare(self):
        self.native.check(self.L.ggml_cdna4_prepare_act(self.t, self.x.data_ptr(), self.k, self.k, self.b, self.ws.data_ptr(), self.ws.numel(), self.ops.PATH_GEMM, self.stream))

    def gemm_only(self):
        self.native.check(self.L.ggml_cdna4_mul_mat_prepared(self.t, self.a.data.data_ptr(), self.a.row_bytes, self.y.data_ptr(), self.m, self.m, self.k, self.b,
                                                             self.ws.data_ptr(), self.ws.numel(), self.ops.PATH_GEMM, self.variant, self.splitk, self.stream))


def events_us(fn, n, warm=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tw = time.perf_counter()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    while time.perf_counter() - tw < 0.05:                            # >= 50 ms of warm-up: the clocks ramp for the first milliseconds
        fn(); fn(); fn(); fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def graph_us(dev, fn, n=40):
    """us per call of fn on the device: n calls captured into a HIP graph on a side stream, the graph replayed under HIP events (no host time between the
    launches: what a decode loop or ggml's graph replay pays).  Falls back to the event loop if the capture fails."""
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        s2 = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s2):
            for _ in range(2):
                fn()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s2):
            for _ in range(n):
                fn()
        return events_us(g.replay, 8, 3) / n
    except Exception:  # noqa: BLE001
        torch.cuda.synchronize(dev)
        return events_us(fn, 100, 10)


def one_launch(t, m, k, b):
    """does ggml_cdna4_mul_mat take this shape as ONE launch (route 11: the activation quantizer and a grid barrier inside k_gemm_kq_t64<.., FQ>)?  Asked of the library's own routing."""
    from ggml_amd import native
    return native.lib().ggml_cdna4_mul_mat_route(t, m, k, b) == 11


def kernel_name(t, m, k, b, fused=False):
    if fused:
        return ("k_gemm_kq_t64<Q4_K, 128, FQ> — ONE launch per step: every work-group quantizes its share of the fp32 activations to the Q8_K-rounded fp16 image (write-through), "
                "grid barrier with the first weight stages already in flight, then the 128x128-tile loop (8 waves x 64(m)x128(b) x K/4, LDS-DMA by the four older waves; split in two by hand-off "
                "inside the resident grid)")
    if t == Q4_K and b > 64:
        nt = ((m + 255) // 256) * ((b + 255) // 256)                   # cdna4_gemm_r8_preferred (gemm_q_lds.hip) on a 256-CU part
        if k % 256 == 0 and nt >= 256 and nt * 10 >= -(-nt // 256) * 256 * 9:
            return "k_gemm_r8<Q4_K> (256x256 tile, 8 waves x 32(m)x256(b), in-register unpack, K tile 64, no K split)"
        t256, t128 = ((m + 255) // 256) * ((b + 127) // 128), ((m + 127) // 128) * ((b + 127) // 128)
        eff = lambda n: n / (-(-n // 256) * 256)                        # cdna4_launch_gemm_t64's tile rule on a 256-CU part
        if t256 * 4 >= 256 * 3 and eff(t256) * 1.10 >= eff(t128):
            return "k_gemm_kq_t64<Q4_K, 256> (256x128 tile, 8 waves x 64(m)x128(b), LDS-DMA by the four older waves in front of the stage barrier, no K split)"
        return "k_gemm_kq_t64<Q4_K, 128> (128x128 tile, 8 waves x 64(m)x128(b) x K/4, LDS-DMA by the four older waves; small grids split in two with the ticketed sum)"
    return "k_gemm_kq_w12 / k_gemm_kq_w8p (128x128 tile, cross-stage unpack/MFMA pipeline)"


# ------------------------------------------------------------------------------------------------ optional legs (rank 0, N = 1)
def cpu_baseline(m, k, b, seconds=14.0):
    """the UNMODIFIED reference CPU backend (oracle/_ref, built by oracle/ref.mk) on the headline workload, host cores of this box:
    a thread sweep over both builds — x86-64-v3 (AVX2: the build the oracle restates) and x86-64-v4 + VNNI (AVX-512: what
    -march=native selects on this host) — best reported, the sweep attached.  Falls back to the C port when the binaries are absent."""
    cores = os.cpu_count() or 1
    sweep, best = [], None
    builds = [("avx512_vnni", os.path.join(REFDIR, "v4", "cpu_baseline")), ("avx2", os.path.join(REFDIR, "cpu_baseline"))]
    builds = [(n, e) for n, e in builds if os.path.exists(e)]
    if builds:
        cands = sorted({c for c in (cores, cores // 2, cores // 4, 64, 32, 16) if 1 <= c <= cores}, reverse=True)
        per = max(0.6, seconds / (len(cands) * len(builds)))
        for name, exe in builds:
            for nt in cands:
                try:
                    out = subprocess.run([exe, "q4_K", str(m), str(k), str(b), str(per), str(nt)], capture_output=True, text=True, timeout=per * 6 + 60)
                    j = json.loads(out.stdout.strip().splitlines()[-1])
                    row = {"build": name, "threads": nt, "tflops": round(j["gflops"] / 1e3, 4), "ms_per_run": round(j["us_per_run"] / 1e3, 3), "runs": j["runs"]}
                    sweep.append(row)
                    if best is None or row["tflops"] > best["tflops"]:
                        best = row
                except Exception as e:  # noqa: BLE001
                    sweep.append({"build": name, "threads": nt, "error": repr(e)[:120]})
        if best:
            return {"value": best["tflops"], "unit": "TFLOP/s", "cores": best["threads"], "host_cores": cores, "kind": "reference", "build": best["build"],
                    "sample": "full workload Q4_K [%dx%d]·[%dx%d], ggml-cpu MUL_MAT (unmodified reference); best of the sweep: %d runs, %.1f ms/run" % (m, k, k, b, best["runs"], best["ms_per_run"]),
                    "thread_sweep": sweep,
                    "note": "the reference's threadpool peaks well below the core count on this shape: 512 activation columns x 4096 rows split into too few chunks for 256 threads (ggml-cpu.c:7560-7590 chunking)"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refutil as R
    w = synth_blocks(Q4_K, 512, k, 7)
    x = np.random.default_rng(8).uniform(-1, 1, (b, k)).astype(np.float32)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        R.o_mul_mat(Q4_K, w, x, 512, k); n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(2.0 * 512 * k * b / dt / 1e12, 4), "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": "512 of %d weight rows x full [%dx%d] activations, oracle C port (OpenMP)" % (m, k, b)}


def library_ceiling(dev, m, k, b, steps):
    """the vendor library's dense fp16 GEMM of the same shape on ALREADY-dequantized weights (torch.matmul -> hipBLASLt): not part
    of the product — the practical ceiling of this shape next to the nominal MFMA roof"""
    try:
        wh = torch.empty((m, k), dtype=torch.float16, device=dev).uniform_(-1, 1)
        xh = torch.empty((b, k), dtype=torch.float16, device=dev).uniform_(-1, 1)
        yh = torch.empty((b, m), dtype=torch.float16, device=dev)
        us = events_us(lambda: torch.matmul(xh, wh.t(), out=yh), steps, 20)
        tf = 2.0 * m * k * b / us / 1e6
        return {"what": "torch.matmul fp16 [%dx%d]·[%dx%d]^T (hipBLASLt) on pre-dequantized weights (%.1f MB of fp16 W instead of %.1f MB of Q4_K)" % (b, k, k, m, m * k * 2 / 1e6, m * k * 0.5625 / 1e6),
                "us_per_launch": round(us, 3), "tflops": round(tf, 2), "frac": round(tf / MFMA_F16_PEAK_TFLOPS, 4)}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def shape_row(dev, t, m, k, b, steps):
    """kernel-only time of the default GEMM route at another shape (activations prepared), same prescribed data"""
    w, x, how = prescribed(t, m, k, 0, m, b)
    h = Hot(dev, t, w, m, k, x)
    h.prepare()
    us = events_us(h.gemm_only, steps, 20)
    step_us = events_us(h.step, steps, 10)
    tf = h.flops / us / 1e6
    row = {"shape": [m, k, b], "gemm_us": round(us, 3), "gemm_tflops": round(tf, 1), "frac": round(tf / MFMA_F16_PEAK_TFLOPS, 4), "step_us": round(step_us, 3),
           "step_tflops": round(h.flops / step_us / 1e6, 1), "kernel": kernel_name(t, m, k, b), "data": how}
    if t == Q4_K and "k_gemm_r8" in row["kernel"]:                      # same box, same data, same minute: the kernel AUTO took until round 3 (variant bit 13 = k_gemm_kq_t64, its own tile choice)
        h.variant = 8192 | 7
        row["gemm_us_k_gemm_kq_t64_same_box"] = round(events_us(h.gemm_only, steps, 20), 3)
        h.variant = 0
        row["gemm_us_again"] = round(events_us(h.gemm_only, steps, 20), 3)
    del h
    lc = library_ceiling(dev, m, k, b, max(10, steps // 2))             # what the vendor library makes of this shape on fp16 weights (is 0.70 attainable here?)
    row["library_ceiling"] = {kk: lc[kk] for kk in ("us_per_launch", "tflops", "frac", "error") if kk in lc}
    return row


def pmc_decode_traffic():
    """HBM-side bytes per launch of the one-launch decode kernel at 4096 x 14336 from the latest committed PMC passes
    (profiles/rNN/pmc_decode_summary.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over scripts/decode_loop.py; FETCH_SIZE in KB, x2 on gfx950)"""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_decode_summary.txt")))
    if not files:
        return None
    rd = wr = None
    for ln in open(files[-1]):
        m = re.search(r"'FETCH_SIZE': ([0-9.]+)", ln)
        if m and "gemv" in ln:
            rd = float(m.group(1)) * 1024 * 2
        m = re.search(r"'WRITE_SIZE': ([0-9.]+)", ln)
        if m and "gemv" in ln:
            wr = float(m.group(1)) * 1024
    return None if rd is None or wr is None else rd + wr


def decode_rows(dev, steps):
    """batch-1 decode (BASELINE configs[1]) and the reference's own perf shape (m = 4096, k = 14336, tests/test-backend-ops.cpp:4340-4346):
    ONE launch per step (activation quantizer fused into the GEMV), 64 rotating copies of W so that every launch streams from HBM"""
    from ggml_amd import native, ops
    L = native.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = {}
    for (m, k) in ((4096, 4096), (4096, 14336)):
        w, x, how = prescribed(Q4_K, m, k, 0, m, 1)
        a = ops.QTensor.from_host_bytes(Q4_K, k, m, w, device=dev)
        ncopy = 64 if k == 4096 else 16                                # 604 MB / 528 MB > 256 MB Infinity Cache
        big = a.data.reshape(-1).repeat(ncopy)
        row_b, mat_b = a.row_bytes, a.row_bytes * m
        x1 = torch.from_numpy(x).to(dev)
        y1 = torch.empty((1, m), dtype=torch.float32, device=dev)
        ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(Q4_K, k, 1), dtype=torch.uint8, device=dev)
        cnt = [0]

        def fused_cold():
            i = cnt[0] % ncopy; cnt[0] += 1
            native.check(L.ggml_cdna4_mul_mat(Q4_K, big.data_ptr() + i * mat_b, row_b, x1.data_ptr(), k, y1.data_ptr(), m, m, k, 1, ws.data_ptr(), ws.numel(), 0, 0, 0, st))

        def fused_warm():
            native.check(L.ggml_cdna4_mul_mat(Q4_K, big.data_ptr(), row_b, x1.data_ptr(), k, y1.data_ptr(), m, m, k, 1, ws.data_ptr(), ws.numel(), 0, 0, 0, st))
        n = max(steps, 256)
        cold, warm = events_us(fused_cold, n, 20), events_us(fused_warm, n, 20)
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(500):
            fused_warm()
        torch.cuda.synchronize(dev); wall = (time.perf_counter() - t0) / 500 * 1e6
        # the same 64 / 16 launches captured once into a HIP graph and replayed (how a decode loop issues them)
        hg = None
        try:
            s2 = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s2):
                st2 = s2.cuda_stream
                def one(i):
                    native.check(L.ggml_cdna4_mul_mat(Q4_K, big.data_ptr() + (i % ncopy) * mat_b, row_b, x1.data_ptr(), k, y1.data_ptr(), m, m, k, 1, ws.data_ptr(), ws.numel(), 0, 0, 0, st2))
                for i in range(ncopy):
                    one(i)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s2):
                st2 = torch.cuda.current_stream(dev).cuda_stream
                for i in range(ncopy):
                    one(i)
            hg = events_us(g.replay, 30, 5) / ncopy
        except Exception as e:  # noqa: BLE001
            hg = repr(e)[:200]
        alg = mat_b + k * 4 + m * 4                                     # W + fp32 x + y (the quantized row never leaves LDS)
        gbs = alg / cold / 1e3
        rows["%dx%d" % (m, k)] = {"us_per_step": round(cold, 3), "us_per_step_cache_warm": round(warm, 3), "us_per_step_host_wall": round(wall, 3),
                                   "us_per_step_hipgraph": round(hg, 3) if isinstance(hg, float) else hg, "tokens_per_s": round(1e6 / cold, 1),
                                   "effective_tflops": round(2.0 * m * k / cold / 1e6, 3),
                                   "roofline": {"bound": "hbm", "kernel": "k_gemv_q_fused<Q4_K>", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                "traffic": pmc_decode_traffic() if k == 14336 else None, "algorithmic_bytes_per_launch": alg}, "data": how}
        del big
    return rows


def format_rows(dev, steps):
    """secondary rows: every weight format at the headline shape — GEMM kernel only and the whole step — and at B = 1"""
    rows = {}
    m, k, b = HEAD
    for t in (12, 13, 14, 2, 8):
        w, x, how = prescribed(t, m, k, 0, m, b)
        h = Hot(dev, t, w, m, k, x)
        h.prepare()
        g = events_us(h.gemm_only, steps, 10)
        s = events_us(h.step, steps, 10)
        h1 = Hot(dev, t, w, m, k, x[:1])
        d = events_us(h1.step, max(steps, 200), 10)
        rows[TYPE_NAME[t]] = {"gemm_b512_us": round(g, 3), "gemm_b512_tflops": round(h.flops / g / 1e6, 1), "step_b512_us": round(s, 3),
                              "decode_b1_us_cache_warm": round(d, 3), "weight_bytes": int(w.size), "data": how}
    return rows


def resident_image_rows(dev, steps):
    """round 5: Q4_0 / Q8_0 / Q6_K prefill with a RESIDENT kernel-native image of the weights (ggml_cdna4_resident_image_*: Q4_0R / Q8_0R re-layouts, Q6_K8 widening — what the
    plug-in's CDNA4_Resident buffer type builds at load) against the per-call route (the staging kernel k_gemm_kq_w12), GEMM only, on one large grid (random valid blocks:
    the reference quantizer would take minutes on 64 M weights); Q4_K / Q5_K read their own 16-byte-aligned blocks and need no image"""
    from ggml_amd import native
    L = native.lib()
    m, k, b = 16384, 4096, 1024
    xs = np.random.default_rng(4321).uniform(-1, 1, (b, k)).astype(np.float32)
    rows = {"shape": [m, k, b], "data": "random-valid-blocks"}
    for t in (12, 2, 8, 14):
        h = Hot(dev, t, synth_blocks(t, m, k, 1234), m, k, xs)
        h.prepare()
        row = {"route": L.ggml_cdna4_mul_mat_route_of(t, h.a.data.data_ptr(), h.a.row_bytes, m, k, b), "gemm_us": round(events_us(h.gemm_only, max(20, steps // 4), 5), 3)}
        n = L.ggml_cdna4_resident_image_size(t, m, k)
        if n:
            img = torch.empty(n, dtype=torch.uint8, device=dev)
            native.check(L.ggml_cdna4_resident_image_register(t, h.a.data.data_ptr(), h.a.row_bytes, m, k, img.data_ptr(), 1, None))
            try:
                row["resident_route"] = L.ggml_cdna4_mul_mat_route_of(t, h.a.data.data_ptr(), h.a.row_bytes, m, k, b)
                row["resident_gemm_us"] = round(events_us(h.gemm_only, max(20, steps // 4), 5), 3)
                row["resident_gemm_tflops"] = round(h.flops / row["resident_gemm_us"] / 1e6, 1)
                row["image_bytes_over_weight_bytes"] = round((n - 256) / float(m * h.a.row_bytes), 3)
            finally:
                L.ggml_cdna4_resident_image_unregister(h.a.data.data_ptr())
        row["gemm_tflops"] = round(h.flops / row["gemm_us"] / 1e6, 1)
        rows[TYPE_NAME[t]] = row
        del h
    return rows


def layer_front_rows(dev, steps, d=4096, b=512, mkv=1024):
    """round 5, through the C-ABI only: the front of a llama-8B-sized attention block — cur = rms_norm(x) * g, then wq (4096 x 4096), wk and wv
    (1024 x 4096: grouped-query) on cur, Q4_K, 512 rows — as a host without the hand-off issues it (op_norm_affine + three ggml_cdna4_mul_mat: three activation
    quantizations) and as the plug-in's graph walk issues it (op_norm_affine_q8_K leaves the image, three ggml_cdna4_mul_mat_prepared: none).  HIP-graph replay, one box,
    alternating; the outputs of the two sequences are compared bit for bit."""
    import ctypes as C
    from ggml_amd import native, ops
    L = native.lib()
    t = Q4_K
    rng = np.random.default_rng(99)
    x = torch.from_numpy(rng.standard_normal((1, 1, b, d)).astype(np.float32)).to(dev)
    g = torch.from_numpy((1 + 0.1 * rng.standard_normal((1, 1, 1, d))).astype(np.float32)).to(dev)
    cur = torch.empty_like(x)
    ws_ = [ops.QTensor.from_host_bytes(t, d, m, synth_blocks(t, m, d, 7 + i), device=dev) for i, m in enumerate((d, mkv, mkv))]
    outs = [[torch.empty((b, w.M), dtype=torch.float32, device=dev) for w in ws_] for _ in range(2)]
    ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(t, d, b), dtype=torch.uint8, device=dev)
    dx, dg, dc = ops._tensor_desc(x, 0), ops._tensor_desc(g, 0), ops._tensor_desc(cur, 0)
    assert L.ggml_cdna4_act_image_key(t, d, d, b) == 19 and L.ggml_cdna4_act_image_key(t, mkv, d, b) == 19

    def st():
        return torch.cuda.current_stream(dev).cuda_stream

    def plain():
        native.check(L.ggml_cdna4_op_norm_affine(C.byref(dx), C.byref(dg), None, C.byref(dc), 1e-5, 1, st()))
        for w, y in zip(ws_, outs[0]):
            native.check(L.ggml_cdna4_mul_mat(t, w.data.data_ptr(), w.row_bytes, cur.data_ptr(), d, y.data_ptr(), w.M, w.M, d, b, ws.data_ptr(), ws.numel(), 0, 0, 0, st()))

    def handed_off():
        native.check(L.ggml_cdna4_op_norm_affine_q8_K(C.byref(dx), C.byref(dg), None, C.byref(dc), 1e-5, 1, t, ws.data_ptr(), ws.numel(), st()))
        for w, y in zip(ws_, outs[1]):
            native.check(L.ggml_cdna4_mul_mat_prepared(t, w.data.data_ptr(), w.row_bytes, y.data_ptr(), w.M, w.M, d, b, ws.data_ptr(), ws.numel(), 0, 0, 0, st()))

    plain(); handed_off(); torch.cuda.synchronize(dev)
    same = all(torch.equal(a.view(torch.int32), c.view(torch.int32)) for a, c in zip(outs[0], outs[1]))
    us = {"plain": [], "handed_off": []}
    for _ in range(2):
        us["plain"].append(graph_us(dev, plain, 20)); us["handed_off"].append(graph_us(dev, handed_off, 20))
    return {"workload": "rms_norm * g -> wq (%dx%d), wk, wv (%dx%d), Q4_K, %d rows; C-ABI, HIP-graph replay" % (d, d, mkv, d, b), "bit_identical": bool(same),
            "launches": {"plain": 1 + 3 * 2 - (1 if one_launch(t, d, d, b) else 0), "handed_off": 4},
            "us_plain": round(min(us["plain"]), 2), "us_handed_off": round(min(us["handed_off"]), 2), "all_us": {k_: [round(v, 2) for v in vs] for k_, vs in us.items()}}


def layer_front_one_row(timeout=100):
    """round 6, through ggml's PUBLIC API on the plug-in (oracle/_ref/split_harness `shared`, HIP-graph replay): a llama-8B-sized attention + FFN front at ONE
    activation row — rms_norm -> {wq, wk, wv + bias}, rms_norm -> {w_gate, w_up} -> w_down; D = 4096, H = 14336 — with the one-row MUL_MATs of one src1 grouped into one launch
    (default) and node by node (GGML_CDNA4_NO_GROUP=1); the FNV-1a of the outputs' bytes must be equal"""
    exe, plugin = os.path.join(REFDIR, "split_harness"), os.path.join(ROOT, "ggml_amd", "lib", "libggml-cdna4.so")
    if not (os.path.exists(exe) and os.path.exists(plugin)):
        return {"error": "oracle/_ref/split_harness or the plug-in is not in the snapshot"}
    out = {"workload": "plug-in, ggml public API: attention + FFN front, D 4096, H 14336, ONE activation row; us per graph (HIP-graph replay of the plug-in)"}
    for t in ("q4_K", "q4_0"):
        row = {}
        for name, env in (("grouped", {}), ("node_by_node", {"GGML_CDNA4_NO_GROUP": "1"})):
            try:
                r = subprocess.run([exe, plugin, t, "4096", "14336", "1", "shared"], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, HARNESS_NO_CPU="1", **env))
                j = json.loads(r.stdout.strip().splitlines()[-1])
                row[name] = {"us_per_graph": j["us_per_graph"], "mul_mats_that_rode_along": j["grouped_first_compute"], "fnv1a": j["fnv1a"]}
            except Exception as e:  # noqa: BLE001
                row[name] = {"error": repr(e)[:200]}
        if all("fnv1a" in v for v in row.values()):
            row["bytes_equal"] = row["grouped"]["fnv1a"] == row["node_by_node"]["fnv1a"]
        out[t] = row
    return out


def batch_sweep(dev, steps):
    """µs per MUL_MAT call (activation quantize included, HIP events) from decode to prefill batch sizes — one-launch GEMV (1..8 rows,
    columns from LDS; 3..8 rows over large matrices: the int8 matrix-core kernel), k_mmq_q4_K for 9..32 rows (mmq_i8.hip: v_mfma_i32_16x16x32_i8),
    then k_gemm_kq_t64 with the deep K split while the grid is far below the chip — at the headline matrix and at the reference's perf
    shape (tests/test-backend-ops.cpp:4340-4346)"""
    from ggml_amd import ops
    out = {}
    for (m, k) in ((4096, 4096), (4096, 14336)):
        w, _, how = prescribed(Q4_K, m, k, 0, m, 1)
        a = ops.QTensor.from_host_bytes(Q4_K, k, m, w, device=dev)
        row = {}
        for b in (1, 2, 3, 4, 8, 16, 32, 64, 128, 256):
            x = torch.from_numpy(np.random.default_rng(b).uniform(-1, 1, (b, k)).astype(np.float32)).to(dev)
            y = torch.empty((b, m), dtype=torch.float32, device=dev)
            row[str(b)] = round(graph_us(dev, lambda: ops.mul_mat(a, x, out=y), 40), 2)
        out["%dx%d" % (m, k)] = {"us_per_call_by_rows": row, "data": how, "method": "HIP-graph replay of 40 calls per point (device time; round 3 timed a Python loop)"}
    return out


def moe_row(dev, steps):
    """MUL_MAT_ID at prefill size: 8 experts x 2 used x 512 tokens, Q4_K experts of 4096 x 4096 — the device-side
    grouping (counting sort of the ids, no host sync) + gather-quantize + ONE grouped GEMM launch; effective flops = 2 M K per
    (token, slot) pair.  Decode (1 token) beside it: one launch."""
    from ggml_amd import native, ops
    L = native.lib()
    n_expert, n_used, n_tok, m, k = 8, 2, 512, 4096, 4096
    w, _, how = prescribed(Q4_K, n_expert * m, k, 0, n_expert * m, 1)
    a = ops.QTensor.from_host_bytes(Q4_K, k, n_expert * m, w, device=dev)
    rng = np.random.default_rng(7)
    out = {"shape": {"n_expert": n_expert, "n_used": n_used, "M": m, "K": k}, "data": how}
    for nt in (n_tok, 1):
        xb = torch.from_numpy(rng.uniform(-1, 1, (nt, n_used, k)).astype(np.float32)).to(dev)
        ids = torch.from_numpy(np.stack([rng.permutation(n_expert)[:n_used] for _ in range(nt)]).astype(np.int32)).to(dev)
        y = torch.empty((nt, n_used, m), dtype=torch.float32, device=dev)
        ws = torch.empty(L.ggml_cdna4_mul_mat_id_workspace_size(Q4_K, k, n_expert, n_used, n_used, nt), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream

        def call():
            native.check(L.ggml_cdna4_mul_mat_id(Q4_K, a.data.data_ptr(), a.row_bytes, m * a.row_bytes, xb.data_ptr(), k, n_used * k, ids.data_ptr(), n_used,
                                                 y.data_ptr(), m, n_used * m, m, k, n_expert, n_used, n_used, nt, ws.data_ptr(), ws.numel(), st))
        us = events_us(call, max(20, steps // 2), 5)
        fl = 2.0 * m * k * nt * n_used
        us_prepared = None
        if nt > 1 and L.ggml_cdna4_mul_mat_id_front_key(Q4_K, a.data.data_ptr(), a.row_bytes, m * a.row_bytes, m, k, n_expert, n_used, n_used, nt, ws.numel()):
            # a second expert stack on the SAME (b, ids) — w_gate behind w_up of a mixture-of-experts layer — multiplies the front the call above left in the workspace: one launch
            def prepared():
                native.check(L.ggml_cdna4_mul_mat_id_prepared(Q4_K, a.data.data_ptr(), a.row_bytes, m * a.row_bytes, xb.data_ptr(), k, n_used * k, ids.data_ptr(), n_used,
                                                              y.data_ptr(), m, n_used * m, m, k, n_expert, n_used, n_used, nt, ws.data_ptr(), ws.numel(), st))
            call(); us_prepared = round(events_us(prepared, max(20, steps // 2), 5), 2)
        if nt > 1:
            out["prefill_512_tokens"] = {"us_per_call": round(us, 2), "effective_tflops": round(fl / us / 1e6, 1), "frac_of_mfma_roof": round(fl / us / 1e6 / MFMA_F16_PEAK_TFLOPS, 4),
                                         "launches": "two: k_moe_sk_front (block 0: stable counting sort of the ids + tile records + work-queue spans; the other blocks: the activation quantizer, token order) "
                                                     "and k_gemm_kq_sk (one persistent work-group per span of (tile, m-tile, superblock) units; partial tiles parked, the last arriver of a tile sums in span order — nobody waits)",
                                         "tiles": "per expert ceil(rows / 128) image tiles; rows are gathered by per-lane source rows, nothing is padded in memory",
                                         "us_per_call_second_stack_on_the_same_front": us_prepared}
        else:
            wbytes = n_used * m * (k // 256) * 144
            out["decode_1_token"] = {"us_per_call": round(us, 2), "GBps": round(wbytes / us / 1e3, 1), "frac_of_hbm_roof": round(wbytes / us / 1e3 / HBM_PEAK_GBS, 4), "launches": "one (k_gemv_q_fused<.., IDS>)"}
    return out


def widening_rows(dev, steps):
    """the rows SURVEY 8(f) ranks after the five formats.  (1) Q5_0 / Q3_K / Q2_K (+ IQ4_NL / IQ4_XS / Q4_1 / Q5_1) at the headline shape: the whole MUL_MAT
    step (activation quantize + per-call exact re-encoding into Q8_0 / Q6_K + that format's MFMA GEMM) and the one-launch decode.
    (2) FLASH_ATTN_EXT, F16 K / V, head size 128, 32 heads, with a mask: prefill rows against the fp16 MFMA roof (4 n_head n_q n_kv hs
    flops), decode rows against the HBM roof (K + V read once)."""
    import ctypes as C
    from ggml_amd import native, ops
    out = {"formats": {}, "flash_attn_ext": {}}
    m, k, b = HEAD
    def fmt(t):
        w, x, how = prescribed(t, m, k, 0, m, b)
        a = ops.QTensor.from_host_bytes(t, k, m, w, device=dev)
        xd = torch.from_numpy(x).to(dev)
        y = torch.empty((b, m), dtype=torch.float32, device=dev)
        s_us = events_us(lambda: ops.mul_mat(a, xd, out=y), steps, 10)
        y1 = torch.empty((1, m), dtype=torch.float32, device=dev)
        x1 = xd[:1].contiguous()
        d_us = events_us(lambda: ops.mul_mat(a, x1, out=y1), max(steps, 200), 10)
        out["formats"][TYPE_NAME[t]] = {"step_b512_us": round(s_us, 3), "step_b512_tflops": round(2.0 * m * k * b / s_us / 1e6, 1), "decode_b1_us_cache_warm": round(d_us, 3),
                                        "weight_bytes": int(w.size), "data": how}
        leg_row("formats", TYPE_NAME[t], out["formats"][TYPE_NAME[t]])
    for t in (6, 11, 10):
        fmt(t)
    rng = np.random.default_rng(11)
    nh = 32
    def pipe_name(hs, n_q, n_kv, masked):               # the launcher's rule (fattn.hip: fa_f16): the tallest query tile that gives every CU a work-group
        for nw in ((8, 4) if hs == 128 else (8, 4, 2)):
            if (n_q + 32 * nw - 1) // (32 * nw) * nh >= 256:
                return "k_flash_attn_pipe<%d, %d, %d>" % (hs, nw, 1 if masked else 0)
        c = 4 if hs == 128 else 2                       # below one work-group per CU: the smallest tile with the keys split from 32 chunks per split on
        it, nc = (n_q + 32 * c - 1) // (32 * c) * nh, (n_kv + 63) // 64
        want = min((256 + it // 2) // it, nc // 32)
        if n_q >= 128 and want >= 2:
            return "k_flash_attn_pipe<%d, %d, %d> in %d key splits + k_flash_attn_pipe_merge" % (hs, c, 1 if masked else 0, -(-nc // -(-nc // want)))
        if hs == 128 and n_kv <= 1024 and (n_q + 127) // 128 * nh >= 128:
            return "k_flash_attn_pipe<128, 4, %d>" % (1 if masked else 0)
        return "k_flash_attn_split<%d>" % hs
    for hs, n_q, n_kv, masked in ((128, 4096, 4096, True), (128, 4096, 4096, "causal"), (128, 4096, 4096, False), (64, 4096, 4096, True), (128, 1024, 1024, True), (128, 512, 512, True), (128, 512, 4096, True), (128, 512, 16384, True), (128, 1, 4096, True), (128, 1, 32768, True)):
        q = torch.from_numpy(rng.uniform(-1, 1, (1, nh, n_q, hs)).astype(np.float32)).to(dev)
        kk = torch.from_numpy(rng.uniform(-1, 1, (1, nh, n_kv, hs)).astype(np.float16)).to(dev)
        vv = torch.from_numpy(rng.uniform(-1, 1, (1, nh, n_kv, hs)).astype(np.float16)).to(dev)
        mk = torch.from_numpy(rng.uniform(-1, 1, ((n_q + 63) // 64 * 64, n_kv)).astype(np.float16)).to(dev)
        if masked == "causal":                                          # what a prefill graph passes: 0 on and below the diagonal, -inf above (half of the key chunks are skipped)
            mk = torch.triu(torch.full((n_q, n_kv), float("-inf"), dtype=torch.float16, device=dev), diagonal=1)
        o = ops.flash_attn_ext(q, kk, vv, mk, 1.0 / np.sqrt(hs))                                   # checks the arguments once; the timed calls go
        dq, dk, dv, dd = (ops._tensor_desc(t_, ty) for t_, ty in ((q, 0), (kk, 1), (vv, 1), (o, 0)))      # straight to the C-ABI (python's part
        dm = ops._tensor_desc(mk.view(1, 1, *mk.shape), 1)                                          # of a call would exceed the decode kernel)
        st, L, sc = torch.cuda.current_stream(dev).cuda_stream, native.lib(), float(1.0 / np.sqrt(hs))
        pm = C.byref(dm) if masked else None
        us = events_us(lambda: native.check(L.ggml_cdna4_op_flash_attn_ext(C.byref(dq), C.byref(dk), C.byref(dv), pm, C.byref(dd), sc, 0.0, 0.0, st)), max(20, steps // 4), 5)
        row = {"us_per_call": round(us, 2)}
        if n_q > 32:
            tf = 4.0 * nh * n_q * n_kv * hs / us / 1e6
            row.update(tflops=round(tf, 1), frac_of_mfma_roof=round(tf / MFMA_F16_PEAK_TFLOPS, 4), kernel=pipe_name(hs, n_q, n_kv, bool(masked)))
        else:
            gb = 4.0 * nh * n_kv * hs / us / 1e3
            row.update(kv_GBps=round(gb, 1), frac_of_hbm_roof=round(gb / HBM_PEAK_GBS, 4), kernel="k_flash_attn_split<128> + k_flash_attn_merge<128>")
        if masked == "causal":
            row.update(note="tflops / frac count the FULL 4 n_head n_q n_kv hs; about half of the key chunks are -inf for their whole query tile and are not walked "
                            "(k_fa_mask_flags + chunk list, bit-identical to walking them)")
        name = "hs%d_h32_q%d_kv%d%s" % (hs, n_q, n_kv, "_causal_mask" if masked == "causal" else ("" if masked else "_no_mask"))
        out["flash_attn_ext"][name] = row
        leg_row("flash_attn_ext", name, row)
    # grouped-query decode (32 query heads on 8 K / V heads, 32 K keys, F16 cache): the heads of a group share a tile — the cache is read once per group (against the HBM roof on the
    # cache bytes) — beside the one-head-per-tile form (CDNA4_FA_NO_PACK=1)
    hs, nkvh, n_kv = 128, 8, 32768
    q = torch.from_numpy(rng.uniform(-1, 1, (1, nh, 1, hs)).astype(np.float32)).to(dev)
    kk = torch.from_numpy(rng.uniform(-1, 1, (1, nkvh, n_kv, hs)).astype(np.float16)).to(dev)
    vv = torch.from_numpy(rng.uniform(-1, 1, (1, nkvh, n_kv, hs)).astype(np.float16)).to(dev)
    mk = torch.from_numpy(rng.uniform(-1, 1, (64, n_kv)).astype(np.float16)).to(dev)
    sc = float(1.0 / np.sqrt(hs))
    row = {}
    for tag, env in (("us_per_call", None), ("us_per_call_one_head_per_tile", "1")):
        if env: os.environ["CDNA4_FA_NO_PACK"] = env
        else: os.environ.pop("CDNA4_FA_NO_PACK", None)
        ops.flash_attn_ext(q, kk, vv, mk, sc)
        row[tag] = round(events_us(lambda: ops.flash_attn_ext(q, kk, vv, mk, sc), max(20, steps // 4), 5), 2)
    os.environ.pop("CDNA4_FA_NO_PACK", None)
    gb = 4.0 * nkvh * n_kv * hs / row["us_per_call"] / 1e3
    row.update(cache_GBps=round(gb, 1), frac_of_hbm_roof=round(gb / HBM_PEAK_GBS, 4), kernel="k_flash_attn_split<128> (four heads per tile) + k_flash_attn_merge<128>", note="includes python's share of a call (a few us)")
    out["flash_attn_ext"]["hs128_h32_kvh8_q1_kv32768_grouped_query"] = row
    leg_row("flash_attn_ext", "hs128_h32_kvh8_q1_kv32768_grouped_query", row)
    # decode over a QUANTIZED KV cache (Q8_0 / Q4_0 rows read by the key-split kernel's dequantizing operand loads): against the HBM roof on the
    # cache bytes actually read (34 / 18 bytes per 32 elements, K + V once)
    def block_rows(bb, nrows):                                        # random block rows: fp16 d = 1/64 in front of bb - 2 random quant bytes per 32 elements
        a = rng.integers(0, 256, (nrows, hs // 32, bb), dtype=np.uint8)
        a[:, :, 0:2] = np.frombuffer(np.float16(1.0 / 64).tobytes(), dtype=np.uint8)
        return a.reshape(nrows, hs // 32 * bb)
    for tname, t, bb in (("q8_0", 8, 34), ("q4_0", 2, 18)):
        hs, n_q, n_kv = 128, 1, 32768
        rb = hs // 32 * bb
        q = torch.from_numpy(rng.uniform(-1, 1, (1, nh, n_q, hs)).astype(np.float32)).to(dev)
        kk = torch.from_numpy(block_rows(bb, nh * n_kv).reshape(1, nh, n_kv, rb)).to(dev)
        vv = torch.from_numpy(block_rows(bb, nh * n_kv).reshape(1, nh, n_kv, rb)).to(dev)
        mk = torch.from_numpy(rng.uniform(-1, 1, (64, n_kv)).astype(np.float16)).to(dev)
        sc = float(1.0 / np.sqrt(hs))
        ops.flash_attn_ext(q, kk, vv, mk, sc, kv_type=t)
        us = events_us(lambda: ops.flash_attn_ext(q, kk, vv, mk, sc, kv_type=t), max(20, steps // 4), 5)
        gb = 2.0 * nh * n_kv * rb / us / 1e3
        row = {"us_per_call": round(us, 2), "cache_GBps": round(gb, 1), "frac_of_hbm_roof": round(gb / HBM_PEAK_GBS, 4), "kernel": "k_flash_attn_split<128, %s> + k_flash_attn_merge<128>" % tname,
               "note": "includes python's share of a call (a few us)"}
        out["flash_attn_ext"]["hs128_h32_q1_kv32768_%s_cache" % tname] = row
        leg_row("flash_attn_ext", "hs128_h32_q1_kv32768_%s_cache" % tname, row)
    for t in (20, 23, 3, 7):               # IQ4_NL / IQ4_XS / Q4_1 / Q5_1: no hardware session of their own before this run (tests/test_gpu_widening.py) — last
        fmt(t)
    return out


def stock_perf_lines(timeout=150):
    """the reference's own perf harness on the plug-in: `test-backend-ops perf -o MUL_MAT -b CDNA40` (unmodified binary, plug-in loaded
    through GGML_BACKEND_PATH) — its q4_K lines at m = 4096, k = 14336 (tests/test-backend-ops.cpp:4340-4346)"""
    exe = os.path.join(REFDIR, "test-backend-ops")
    plugin = os.path.join(ROOT, "ggml_amd", "lib", "libggml-cdna4.so")
    if not (os.path.exists(exe) and os.path.exists(plugin)):
        return {"error": "oracle/_ref/test-backend-ops or the plug-in is not in the snapshot"}
    def run(backend, tmo, env):
        try:
            r = subprocess.run([exe, "perf", "-o", "MUL_MAT", "-b", backend], capture_output=True, text=True, timeout=tmo, env=env)
            txt = r.stdout
        except subprocess.TimeoutExpired as e:
            txt = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        return [ln.strip() for ln in txt.splitlines() if "type_a=q4_K" in ln and "not supported" not in ln and "us/run" in ln][:8]      # (a line cut off by the time budget has no figures)
    out = {"command": "GGML_BACKEND_PATH=libggml-cdna4.so test-backend-ops perf -o MUL_MAT -b CDNA40 | -b CPU  (unmodified binary; its q4_K lines, m = 4096, k = 14336)",
           "q4_K": run("CDNA40", timeout, dict(os.environ, GGML_BACKEND_PATH=plugin))}
    # the same harness on the reference CPU backend of this box (all host cores, the harness's default): as many q4_K lines as fit the budget
    cpu_budget = int(max(0, min(90, time_left(420))))
    out["q4_K_cpu_backend"] = (run("CPU", cpu_budget, dict(os.environ)) or "no q4_K line finished within %d s (the harness runs every type at every n, all host cores)" % cpu_budget) if cpu_budget >= 20 else "skipped: time budget"
    return out


# ------------------------------------------------------------------------------------------------ multi-GPU legs
def c5_leg(dist, dev, rank, world, steps, warmup):
    """BASELINE configs[4]: Q4_K [32768x8192]·[8192x512], strong scaling over the ranks (see module doc)"""
    M, K, B = C5
    lo, hi = M * rank // world, M * (rank + 1) // world
    w, x, how = prescribed(Q4_K, M, K, lo, hi, B)
    h = Hot(dev, Q4_K, w, hi - lo, K, x)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn):
        tw = time.perf_counter()
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        while time.perf_counter() - tw < 0.15:                        # these steps are long and few: let the clocks settle (>= 150 ms of warm-up)
            fn(); torch.cuda.synchronize(dev)
        barrier(); t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier(); el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el / steps * 1e3
    flops = 2.0 * M * K * B
    out = {"workload": "Q4_K [32768x8192]·[8192x512] row-sharded over %d GPU(s), strong scaling" % world, "data": how}
    ms = timed(h.step)
    out["compute_only"] = {"ms_per_step": round(ms, 5), "tflops": round(flops / ms / 1e9, 2), "frac_of_n_gpu_roof": round(flops / ms / 1e9 / (MFMA_F16_PEAK_TFLOPS * world), 4),
                           "what": "each rank: quantize + GEMM over its %d rows; output left sharded (the consumer of a row-split layer reads its own shard)" % (hi - lo)}
    if dist is not None:
        yfull = torch.empty((world, B, hi - lo), dtype=torch.float32, device=dev)
        ms = timed(lambda: (h.step(), dist.all_gather_into_tensor(yfull, h.y)))
        out["with_allgather_fp32"] = {"ms_per_step": round(ms, 5), "tflops": round(flops / ms / 1e9, 2), "collective": "RCCL all_gather_into_tensor, %d B/rank" % (B * (hi - lo) * 4)}
        y16 = torch.empty((B, hi - lo), dtype=torch.float16, device=dev)
        yfull16 = torch.empty((world, B, hi - lo), dtype=torch.float16, device=dev)
        ms = timed(lambda: (h.step(), y16.copy_(h.y), dist.all_gather_into_tensor(yfull16, y16)))
        out["with_allgather_fp16"] = {"ms_per_step": round(ms, 5), "tflops": round(flops / ms / 1e9, 2), "collective": "fp32 -> fp16 cast + RCCL all_gather_into_tensor, %d B/rank" % (B * (hi - lo) * 2)}
        # K-split: rank r owns superblocks [r K/N, (r+1) K/N) of EVERY row.  A K range of a block-quantized row is a byte range of
        # that row (144 B per 256 weights), so the shard is a strided view of the full matrix: built here from the prescribed matrix.
        sys.path.insert(0, ROOT)
        from ggml_amd import shard as SH
        wfull, _, _ = prescribed(Q4_K, M, K, 0, M, B)
        wk, klo, khi = SH.k_shard_bytes(wfull, M, K, 256, 144, rank, world)
        del wfull
        hk = Hot(dev, Q4_K, wk, M, khi - klo, x[:, klo:khi])
        ms = timed(lambda: (hk.step(), dist.all_reduce(hk.y)))
        # correctness: the all-reduced K-split result against the all-gathered row-split result (same matrix, same activations)
        h.step(); dist.all_gather_into_tensor(yfull, h.y)
        hk.step(); dist.all_reduce(hk.y)
        torch.cuda.synchronize(dev)
        yrow = torch.cat([yfull[r] for r in range(world)], dim=1).double()
        err = float((hk.y.double() - yrow).norm() / yrow.norm())
        out["ksplit_allreduce"] = {"ms_per_step": round(ms, 5), "tflops": round(flops / ms / 1e9, 2), "collective": "RCCL all_reduce(sum) of the fp32 partial outputs, %d B" % (B * M * 4),
                                   "rel_l2_vs_row_split": err, "check": "pass" if err < 2e-3 else "FAIL"}
        out["accounting"] = ("compute_only is the number the >= 6x target of the north star can refer to (output left sharded); gathering the fp32 output moves "
                             "%.1f MB per rank over xGMI (>= %.0f us at 153 GB/s per link), the fp16 gather half of that" % (B * (hi - lo) * 4 / 1e6, B * (hi - lo) * 4 / 153e3))
    return out


def leg_in_child(name, steps, timeout):
    """one optional leg in a process of its own (python bench.py --leg NAME): its result — the rows that were finished when it ended, if it
    was cut short (the child prints every row as soon as it has it) — plus what went wrong, if anything did"""
    note, txt, rc = None, "", 0
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name, "--steps", str(steps)], capture_output=True, text=True, timeout=timeout)
        txt, rc = r.stdout, r.returncode
        if rc:
            note = "leg '%s' exited with %d after the rows below: %s" % (name, rc, (r.stderr or "")[-300:])
    except subprocess.TimeoutExpired as e:
        txt = (e.stdout.decode() if isinstance(e.stdout, bytes) else e.stdout) or ""
        note = "leg '%s' was stopped after %d s; rows finished by then are below" % (name, timeout)
    out = {}
    for ln in txt.splitlines():
        if ln.startswith("LEG_ROW "):
            _, group, key, body = ln.split(" ", 3)
            out.setdefault(group, {})[key] = json.loads(body)
    if note:
        out["error"] = note
    return out


def leg_row(group, key, row):
    """(child side) one finished row, printed at once"""
    print("LEG_ROW %s %s %s" % (group, key, json.dumps(row)), flush=True)


def run_legs(legs, out):
    """the optional legs (name, thunk, latest start in seconds after the run began): none STARTS once the run is that old, and a leg
    that raises records the error instead of taking the metric line down"""
    for name, fn, budget in legs:
        if time_left(budget) <= 0:
            out[name] = "skipped: time budget"
            continue
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="headline", choices=["headline", "c5"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the metric, its roofline, the library ceiling and the CPU baseline")
    ap.add_argument("--no-diagnostics", action="store_true", help="(accepted for compatibility; same as --no-extras)")
    ap.add_argument("--leg", default=None, help="(internal) run ONE optional leg in this process and print its JSON: the parent runs legs that touch "
                                                "paths without a hardware session of their own this way, so that a device fault there cannot take the metric line down")
    ap.add_argument("--lean", action="store_true", help="profiling runs (scripts/gpu_full.sh): the timed loop and the kernel-only loop, nothing else — every launch of the "
                                                        "dominant kernel in the trace is a headline launch on the prescribed data")
    args = ap.parse_args()
    extras = not (args.no_extras or args.no_diagnostics or args.lean)
    if args.lean:
        args.no_cpu_baseline = True

    # `python bench.py --gpus N` without a launcher (a driver that only adds --gpus 8 to the 1-GPU command used to get a world of ONE): N ranks are
    # started here, under the launcher the driver itself would use; rank 0's JSON line is this process's stdout
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("BENCH_NO_SPAWN") != "1":
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
    # bench.py OWNS its device (one process per GPU, nothing else on it): it opts in to the routes that wait for co-resident work-groups (round 6: the library's DEFAULT is
    # the shared mode, in which no launch ever waits for another work-group) and says so in `config`; the default mode's step time is reported beside it
    os.environ.setdefault("GGML_CDNA4_OWNED_DEVICE", "1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    selftest = os.environ.get("BENCH_DIST_SELFTEST") == "1"                  # tests/test_bench_host.py: the launch / rendezvous / reduce / rank-0 plumbing without a GPU (gloo)
    if args.gpus > 1 or world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo" if selftest and not torch.cuda.is_available() else "nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    if selftest:
        # no GPU work: K "steps" of host time, the barrier + max-over-ranks reduction of the real path, ONE line from rank 0 with the world's size
        if dist is not None: dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps): time.sleep(0.0005 * (rank + 1))
        if dist is not None: dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if dist is not None: dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "selftest (no GPU work)", "value": None, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(float(el.item()) / max(1, args.steps) * 1e3, 5), "selftest": True, "gpus_requested": args.gpus}), flush=True)
        if dist is not None: dist.barrier(); dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from ggml_amd import native
    native.lib()

    if args.leg is not None:
        if args.leg == "layer_front":
            print(json.dumps({"layer_front": layer_front_rows(dev, max(50, min(args.steps, 200)))}), flush=True)
            return
        if args.leg == "resident_images":                                # (sessions short of GPU time measure this leg alone)
            print(json.dumps({"resident_images": resident_image_rows(dev, max(50, min(args.steps, 200)))}), flush=True)
            return
        if args.leg == "mul_mat_id":
            print(json.dumps({"mul_mat_id": moe_row(dev, max(50, min(args.steps, 200)))}), flush=True)
            return
        fn = {"widening": lambda: widening_rows(dev, max(50, min(args.steps, 200)))}[args.leg]
        fn()                                                             # prints its rows itself (leg_row)
        return

    if args.config == "c5":
        c5 = c5_leg(dist, dev, rank, world, args.steps, args.warmup)
        if rank == 0:
            ms = c5["compute_only"]["ms_per_step"]
            print(json.dumps({"metric": "effective TFLOPS (2*M*N*K), Q4_K mul_mat [32768x8192]x[8192x512] row-sharded", "value": c5["compute_only"]["tflops"], "unit": "TFLOP/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                              "dtype": "f16", "data": "synthetic (%s)" % c5["data"], "config": {"workload": c5["workload"], "M": C5[0], "K": C5[1], "B": C5[2], "parallelism": "row-split x%d" % world},
                              "tokens_per_s": round(C5[2] / (ms * 1e-3), 1), "c5": c5}), flush=True)
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    # ---- headline: this rank's row shard of the [4096*world x 4096] matrix, the (replicated) activations
    M, K, B = HEAD
    w, x, how = prescribed(Q4_K, M * world, K, M * rank, M * (rank + 1), B)
    h = Hot(dev, Q4_K, w, M, K, x, args.variant, args.splitk)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        h.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    flops_step = h.flops * world
    value = flops_step / (ms_per_step * 1e-3) / 1e12

    # ---- the dominant kernel alone, HIP events on the launch stream.  Where the step is ONE launch (route 11) that launch IS the dominant kernel and its duration is the
    # step's; the two launches it replaces (activation quantizer alone, GEMM alone on prepared activations) are timed beside it for the record
    fused = args.variant == 0 and args.splitk == 0 and one_launch(Q4_K, M, K, B)
    quant_us = events_us(h.prepare, args.steps, 10)
    h.prepare()
    gemm2_us = events_us(h.gemm_only, args.steps, 10)
    gemm_us = events_us(h.step, args.steps, 10) if fused else gemm2_us
    gemm_tf = h.flops / gemm_us / 1e6
    out = None
    if rank == 0:
        # the same launches on all-zero operands: identical instruction stream and traffic, no toggling in the matrix pipe — the gap
        # is what the chip's power management takes (a diagnostic beside the roofline fraction, not a result)
        zero_us = None
        try:
            if args.lean:
                raise RuntimeError("lean")
            hz = Hot(dev, Q4_K, np.zeros_like(w), M, K, np.zeros_like(x), args.variant, args.splitk)
            hz.ws.zero_()
            zero_us = round(events_us(hz.step if fused else hz.gemm_only, 40, 10), 3)
            del hz
        except Exception:  # noqa: BLE001
            pass
        out = {
            "metric": "effective TFLOPS (2*M*N*K), Q4_K mul_mat [4096x4096]x[4096x512]", "value": round(value, 3), "unit": "TFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic (%s)" % how,
            "config": {"workload": "Q4_K MUL_MAT [4096x4096]·[4096x512] per GPU; step = Q8_K activation quantize + fp16-MFMA GEMM, W and fp32 X resident in HBM",
                       "M_per_gpu": M, "K": K, "B": B, "parallelism": "row-split x%d, output left sharded" % world, "gemm_variant": args.variant, "splitk": args.splitk,
                       "device_mode": "owned (GGML_CDNA4_OWNED_DEVICE=1: bench.py has the GPU to itself); the library's default is the shared mode — see roofline.step_us_default_shared_mode"},
            "tokens_per_s": round(B * world / (ms_per_step * 1e-3), 1),
            "roofline": {"bound": "mfma", "kernel": kernel_name(Q4_K, M, K, B, fused) if args.variant == 0 else "gemm variant %d" % args.variant,
                         "achieved": round(gemm_tf, 3), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gemm_tf / MFMA_F16_PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic("k_gemm_kq_t64<12, 128, false, 0, false, true>" if fused else "k_gemm_kq_t64<12, 128, false, 0, false, false>") if args.variant == 0 else None,
                         "traffic_note": "HBM-side bytes/launch (FETCH_SIZE x2 + WRITE_SIZE) from the rocprofv3 PMC passes of this command, profiles/rNN/pmc_summary.txt",
                         "us_per_launch": round(gemm_us, 3), "algorithmic_flops_per_launch": h.flops, "us_per_launch_all_zero_operands": zero_us,
                         # the whole step as the host loop times it (one launch, or quantizer + GEMM + the gap between them) against the same roof: value / peak
                         "step_frac": round(value / world / MFMA_F16_PEAK_TFLOPS, 4),
                         "launches_per_step": 1 if fused else 2,
                         # the two launches the one-launch step replaces, each alone: the activation quantizer and the GEMM on prepared activations (AUTO's ticketed split)
                         "two_launch_components_us": {"k_quantize_q8_K": round(quant_us, 3), "k_gemm_kq_t64<Q4_K, 128>": round(gemm2_us, 3)},
                         # the in-graph case: the producer of src1 — the NORM chain's launch — already left the activation image, so the MUL_MAT node is the GEMM
                         # alone on prepared activations (ggml_cdna4_mul_mat_prepared: what the plug-in issues for wq / wk / wv / w_gate / w_up behind a norm)
                         "step_us_image_from_producer": round(gemm2_us, 3), "step_frac_image_from_producer": round(h.flops / gemm2_us / 1e6 / MFMA_F16_PEAK_TFLOPS, 4)},
        }
        # the same step in the library's DEFAULT mode (shared device: quantizer launch + GEMM with the ticketed split — nothing waits for a co-resident work-group)
        try:
            L_ = native.lib(); old_mode = L_.ggml_cdna4_set_shared_device(1)
            out["roofline"]["step_us_default_shared_mode"] = round(events_us(h.step, args.steps, 10), 3)
            L_.ggml_cdna4_set_shared_device(old_mode)
        except Exception as e:  # noqa: BLE001
            out["roofline"]["step_us_default_shared_mode"] = repr(e)[:120]
        # the step's other kernel, HBM-bound: reads the fp32 activations once, writes the pair-interleaved fp16 image + per-256 scale and per-16 sums
        qbytes = B * K * 4 + B * K * 2 + B * (K // 256) * 4 + B * (K // 16) * 2
        out["quantizer"] = {"kernel": "k_quantize_q8_K (fp32 rows -> fp16 GEMM image of the Q8_K-rounded activations, d, bsums)", "bound": "hbm", "us_per_launch": round(quant_us, 3),
                            "algorithmic_bytes_per_launch": qbytes, "achieved": round(qbytes / quant_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(qbytes / quant_us / 1e3 / HBM_PEAK_GBS, 4)}
        if world == 1 and not args.lean:
            out["roofline"]["library_ceiling"] = library_ceiling(dev, M, K, B, args.steps)
            lc = out["roofline"]["library_ceiling"]
            if "tflops" in lc:
                lc["ours_over_library"] = round(gemm_tf / lc["tflops"], 3)

    if dist is not None:
        # the headline's own collective legs (`value` is the compute-only weak-scaling number — the output of a row-split layer stays sharded, its
        # consumer reads its own shard — and a >= 6x there says nothing about xGMI): the same step followed by the RCCL all-gather of the output, fp32 and fp16
        legs = {"compute_only": {"ms_per_step": round(ms_per_step, 5), "tflops": round(value, 3)}}
        yfull = torch.empty((world, B, M), dtype=torch.float32, device=dev)
        y16 = torch.empty((B, M), dtype=torch.float16, device=dev)
        yfull16 = torch.empty((world, B, M), dtype=torch.float16, device=dev)
        for name, fn, nbytes in (("with_allgather_fp32", lambda: (h.step(), dist.all_gather_into_tensor(yfull, h.y)), B * M * 4),
                                 ("with_allgather_fp16", lambda: (h.step(), y16.copy_(h.y), dist.all_gather_into_tensor(yfull16, y16)), B * M * 2)):
            for _ in range(args.warmup): fn()
            barrier(); t1 = time.perf_counter()
            for _ in range(args.steps): fn()
            barrier(); el = time.perf_counter() - t1
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms_c = float(tt.item()) / args.steps * 1e3
            legs[name] = {"ms_per_step": round(ms_c, 5), "tflops": round(flops_step / (ms_c * 1e-3) / 1e12, 3), "collective": "RCCL all_gather_into_tensor, %d B/rank" % nbytes}
        c5 = c5_leg(dist, dev, rank, world, max(10, args.steps // 4), max(3, args.warmup // 4))      # a default multi-GPU run also carries BASELINE configs[4]
        if rank == 0:
            out["collective_legs"] = legs
            out["accounting"] = ("`value` = compute_only: weak scaling of the headline with the output left sharded (no collective in the timed region); collective_legs has the same step "
                                 "with the RCCL all-gather of the output behind it; c5 (BASELINE configs[4], strong scaling) carries compute_only / with_allgather_* / ksplit_allreduce")
            out["c5"] = c5

    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, K, B)
        if extras:
            steps = max(50, min(args.steps, 200))
            legs = (("decode", lambda: decode_rows(dev, steps), 150),
                    ("shapes", lambda: {"c3_4096x11008x512": shape_row(dev, Q4_K, 4096, 11008, 512, steps), "c5_32768x8192x512_one_gpu": shape_row(dev, Q4_K, 32768, 8192, 512, max(20, steps // 4))}, 200),
                    ("formats", lambda: format_rows(dev, steps), 230),
                    ("resident_images", lambda: resident_image_rows(dev, steps), 238),
                    ("layer_front", lambda: layer_front_rows(dev, steps), 242),
                    ("layer_front_one_row", lambda: layer_front_one_row(), 244),
                    ("mul_mat_id", lambda: moe_row(dev, steps), 245),
                    ("batch_sweep", lambda: batch_sweep(dev, steps), 250),
                    ("widening", lambda: leg_in_child("widening", steps, int(max(30, min(120, time_left(330))))), 255),
                    ("stock_test_backend_ops_perf", lambda: stock_perf_lines(int(max(30, min(150, time_left(400))))), 280))
            run_legs(legs, out)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
