"""The N > 1 path on CPU: world_size-2 gloo processes run the row-split partition (ggml_amd.shard) with the CPU
oracle standing in for the kernel, gather the output shards and compare with the unsharded result (bit-exact:
rows are independent).  Also the pure rank arithmetic."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import refutil as R
from ggml_amd import shard


def test_row_ranges_cover_and_align():
    for M, world in [(4096, 1), (4096, 2), (4096, 8), (32768, 8), (1000, 3), (130, 4)]:
        rs = shard.all_row_ranges(M, world)
        assert rs[0][0] == 0 and rs[-1][1] == M
        for (a, b), (c, d) in zip(rs, rs[1:]):
            assert b == c and a <= b
        for lo, hi in rs[:-1]:
            assert lo % shard.ROW_TILE == 0 and hi % shard.ROW_TILE == 0
    assert shard.all_row_ranges(32768, 8) == [(4096 * r, 4096 * (r + 1)) for r in range(8)]
    with pytest.raises(ValueError):
        shard.row_range(10, 2, 2)


def _worker(rank, world, port, M, K, B, t, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        w = R.random_block_bytes(t, M, K, rng)
        x = np.random.default_rng(6).uniform(-1, 1, (B, K)).astype(np.float32)
        lo, hi = shard.row_range(M, rank, world)
        rs = R.row_size(t, K)
        y_local = torch.from_numpy(R.o_mul_mat(t, w[lo * rs:hi * rs], x, hi - lo, K))
        y = shard.gather_rows(y_local, M, world)
        if rank == 0:
            full = R.o_mul_mat(t, w, x, M, K)
            q.put(bool(np.array_equal(y.numpy(), full)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M", [512, 384])        # equal shards (one all_gather_into_tensor) and ragged shards
def test_rowsplit_gather_world2(M):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + M) % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, M, 512, 5, R.Q4_K, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_k_ranges_cover_whole_superblocks():
    for K, world in [(8192, 8), (8192, 1), (4096, 3), (11008, 2), (11008, 8), (256, 1)]:
        rs = [shard.k_range(K, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == K
        for (a, b), (c, d) in zip(rs, rs[1:]):
            assert b == c
        assert all(lo % 256 == 0 and hi % 256 == 0 for lo, hi in rs)
    assert shard.k_range(11008, 0, 2) == (0, 22 * 256) and shard.k_range(11008, 1, 2) == (22 * 256, 43 * 256)


def _worker_ksplit(rank, world, port, M, K, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = R.random_block_bytes(R.Q4_K, M, K, np.random.default_rng(5))
        x = np.random.default_rng(6).uniform(-1, 1, (B, K)).astype(np.float32)
        wk, lo, hi = shard.k_shard_bytes(w, M, K, 256, 144, rank, world)
        # the oracle stands in for the kernel: a partial product over this rank's K range (activation superblocks are
        # quantized independently, so the partials of the K ranges add up to the full product's superblock terms)
        y = torch.from_numpy(R.o_mul_mat(R.Q4_K, wk, np.ascontiguousarray(x[:, lo:hi]), M, hi - lo))
        dist.all_reduce(y)
        if rank == 0:
            full = R.o_mul_mat(R.Q4_K, w, x, M, K)
            q.put(float(R.rel_l2(y.numpy(), full)))
    finally:
        dist.destroy_process_group()


def test_ksplit_allreduce_world2():
    """the K-split variant (bench.py config c5, `ksplit_allreduce`): per-rank partial products over whole-superblock K ranges,
    summed by an all-reduce, equal the unsplit product up to fp32 summation order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 300
    procs = [ctx.Process(target=_worker_ksplit, args=(r, 2, port, 256, 1792, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=10) < 1e-6
