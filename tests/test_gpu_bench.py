"""-m gpu: bench.py's multi-GPU legs execute on one GPU — a RCCL world of one (BENCH_FORCE_DIST=1): the row-split strong-scaling run of BASELINE configs[4]
([32768 x 8192] . [8192 x 512]) with its all-gather and K-split + all-reduce variants and their on-GPU correctness check, so that the code the driver
launches at N = 2 / 4 / 8 has at least run."""
import json
import os
import subprocess
import sys

import pytest

import refutil as R

pytestmark = pytest.mark.gpu


def test_c5_legs_run_in_a_world_of_one():
    env = dict(os.environ, BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(R.ROOT, "bench.py"), "--config", "c5", "--steps", "6", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert js, (r.stdout + r.stderr)[-3000:]
    d = json.loads(js[-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 100, d
    c5 = d["c5"]
    for leg in ("compute_only", "with_allgather_fp32", "with_allgather_fp16", "ksplit_allreduce"):
        assert leg in c5 and c5[leg].get("ms_per_step", 0) > 0, (leg, c5.get(leg))
    assert c5["ksplit_allreduce"].get("check") == "pass", c5["ksplit_allreduce"]
