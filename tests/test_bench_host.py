"""Host-side logic of bench.py that can be checked without a GPU: the optional legs' start deadlines (a leg that would start
after its deadline says so instead of running; a leg that raises records the error), the prescribed-input generator (the
reference's own mt19937(1234) + quantize_chunk, oracle/_ref/synth_data) sharding consistently, and the kernel the roofline names."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


def test_optional_legs_respect_their_start_deadlines():
    b = _bench()
    ran = []
    legs = (("a", lambda: ran.append("a") or {"ok": 1}, 1e9), ("late", lambda: ran.append("late"), -1000), ("boom", lambda: 1 / 0, 1e9))
    out = b.run_legs(legs, {})
    json.dumps(out)
    assert ran == ["a"] and out["a"] == {"ok": 1} and out["late"] == "skipped: time budget" and "ZeroDivisionError" in out["boom"]["error"]


def test_prescribed_inputs_shard_consistently():
    """rows [lo, hi) of the prescribed matrix are the same bytes whether generated alone or as part of the whole (what lets every
    rank of a row-split run build only its own shard), and the activations do not depend on the shard"""
    b = _bench()
    if not os.path.exists(os.path.join(b.REFDIR, "synth_data")):
        pytest.skip("oracle/_ref/synth_data not built")
    w_all, x_all, how = b.prescribed(b.Q4_K, 512, 1024, 0, 512, 8)
    w_hi, x_hi, _ = b.prescribed(b.Q4_K, 512, 1024, 256, 512, 8)
    assert how == "prescribed" and w_all.size == 512 * 1024 // 256 * 144
    assert np.array_equal(w_all[w_all.size // 2:], w_hi) and np.array_equal(x_all, x_hi)
    assert np.abs(x_all).max() <= 1.0 and len(np.unique(w_all)) > 200


def test_roofline_names_the_kernel_the_auto_route_launches():
    b = _bench()
    assert "k_gemm_kq_t64<Q4_K, 128>" in b.kernel_name(b.Q4_K, *b.HEAD)       # 32 x 4 tiles: the split-K 128x128 form
    assert "k_gemm_r8<Q4_K>" in b.kernel_name(b.Q4_K, *b.C5)                   # 128 x 2 tiles of 256 x 256 = one per CU, unsplit (round 4)
    assert "k_gemm_kq_t64<Q4_K" in b.kernel_name(b.Q4_K, 16384, 8192, 512)      # 128 tiles: r8 would need its split-K exchange, level with t64 at best
    assert "k_gemm_kq_t64<Q4_K, 128>" in b.kernel_name(b.Q4_K, 4096, 11008, 512)      # C3: 32 tiles of 256 x 256 would need an 8-way exchange
    assert "k_gemm_kq_t64<Q4_K, 256>" in b.kernel_name(b.Q4_K, 24576, 8192, 1024)     # 96 x 4 = 384 tiles of 256 x 256: 1.5 rounds, not preferred


def test_layer_front_leg_on_the_emulator():
    """bench.py's layer_front leg (C-ABI only: rms_norm + three products with three activation quantizations vs the NORM-produced image + three prepared products) at a
    small size against the whole-library CPU emulation: the two sequences run, their outputs are bit-identical, the row has the keys the JSON line carries"""
    import json
    import subprocess
    import sys
    import pytest
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: bench.py runs on it")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, sys, os
sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, %r)
import emul_torch
emul_torch.activate()
import torch, bench
bench.graph_us = lambda dev, fn, n=40: (fn(), 1.0)[1]
print(json.dumps(bench.layer_front_rows("cuda:0", 50, d=512, b=96, mkv=256)))
""" % (root, root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, cwd=root)
    if "cannot host the emulation" in (r.stdout + r.stderr):
        pytest.skip("the environment cannot host the emulation")
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    row = json.loads(r.stdout.strip().splitlines()[-1])
    assert row["bit_identical"] is True and row["launches"]["handed_off"] == 4 and "us_plain" in row and "us_handed_off" in row


def test_gpus_n_without_a_launcher_spawns_n_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (what a driver that only adds --gpus N to the 1-GPU command runs) re-executes itself under torch.distributed.run:
    two ranks rendezvous on 127.0.0.1, the max-over-ranks reduction runs, and rank 0 alone prints ONE line that says n_gpus = 2.  BENCH_DIST_SELFTEST=1 replaces the GPU
    work by host sleeps (gloo instead of RCCL): the plumbing is what is under test."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_NO_SPAWN")}
    env["BENCH_DIST_SELFTEST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["gpus_requested"] == 2 and lines[0]["steps"] == 5, lines
    assert lines[0]["ms_per_step"] >= 0.9                              # the MAX over the ranks: rank 1 sleeps 1 ms per step, rank 0 half of that
    # ... and under a launcher (WORLD_SIZE set) it does not spawn again: a world of one stays one process
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    l1 = [json.loads(l) for l in r1.stdout.splitlines() if l.startswith("{")]
    assert r1.returncode == 0 and len(l1) == 1 and l1[0]["n_gpus"] == 1, (r1.stdout + r1.stderr)[-1000:]
