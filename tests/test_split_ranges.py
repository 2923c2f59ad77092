"""CPU: the partition arithmetic of the plug-in's split buffer types (ggml_backend_split_buffer_type — the reference's row split, src/ggml-cuda/ggml-cuda.cu:729-742 — and
ggml_backend_cdna4_ksplit_buffer_type), through the exported ggml_backend_cdna4_split_ranges: the same shard_rows / shard_k that init_tensor calls.  No multi-GPU node has been
in any round's pool, so what can be pinned without one is pinned here: main_device != 0, uneven and zero shares, ragged row counts and K."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "ggml_amd", "lib", "libggml-cdna4.so")
BASE = os.path.join(ROOT, "oracle", "_ref", "libggml-base.so")


@pytest.fixture(scope="module")
def ranges():
    if not (os.path.exists(PLUGIN) and os.path.exists(BASE)):
        pytest.skip("needs the built plug-in and oracle/_ref/libggml-base.so (reference tree present at build time)")
    ctypes.CDLL(BASE, mode=ctypes.RTLD_GLOBAL)                       # the vtable helpers the plug-in links against
    lib = ctypes.CDLL(PLUGIN)
    f = lib.ggml_backend_cdna4_split_ranges
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]

    def call(ksplit, main, ndev, split, n):
        lo, hi, dev = (ctypes.c_int64 * 16)(), (ctypes.c_int64 * 16)(), (ctypes.c_int * 16)()
        ts = None if split is None else (ctypes.c_float * 16)(*split)
        r = f(ksplit, main, ndev, ts, n, lo, hi, dev)
        return r, list(lo[:max(r, 0)]), list(hi[:max(r, 0)]), list(dev[:max(r, 0)])
    return call


def _is_partition(lo, hi, n):
    assert lo[0] == 0 and hi[-1] == n
    for i in range(len(lo)):
        assert lo[i] <= hi[i]
        if i:
            assert lo[i] == hi[i - 1]


@pytest.mark.parametrize("main", [0, 3, 7])
@pytest.mark.parametrize("n", [32768, 4096, 4100, 50257, 130, 0])
def test_row_split_is_a_partition_on_tile_boundaries_for_any_main_device(ranges, main, n):
    for split in (None, [1] * 8, [3, 1, 1, 1, 0, 0, 1, 1], [0, 0, 0, 0, 0, 0, 0, 5], [0.1, 0.2, 0.3, 0.05, 0.05, 0.1, 0.1, 0.1]):
        r, lo, hi, dev = ranges(0, main, 8, split, n)
        assert r == 8 and dev == list(range(8))                      # shard i lives on device i whichever device is the main one
        _is_partition(lo, hi, n)
        assert all(e % 128 == 0 for e in hi[:-1])                    # inner edges on the 128-row GEMM tile
        if split is None or len(set(split)) == 1:                    # equal shares of a tile-aligned matrix: equal shards
            if n % (128 * 8) == 0:
                assert all(h - l == n // 8 for l, h in zip(lo, hi))
        if split == [0, 0, 0, 0, 0, 0, 0, 5]:
            assert hi[6] == 0 and lo[7] == 0 and hi[7] == n          # everything on the one device with a share


@pytest.mark.parametrize("main", [0, 5])
@pytest.mark.parametrize("K", [8192, 4096, 11008, 256, 4352, 4128])
def test_k_split_shards_whole_superblocks_and_leaves_the_tail_to_the_last(ranges, main, K):
    for ndev, split in ((8, None), (8, [2, 1, 1, 1, 1, 1, 1, 0]), (2, [1, 3]), (3, None)):
        r, lo, hi, dev = ranges(1, main if main < ndev else 0, ndev, split, K)
        assert r == ndev
        _is_partition(lo, hi, K)
        assert all(e % 256 == 0 for e in hi[:-1])                    # whole 256-weight superblocks per shard; a ragged K (4128 = 16 x 256 + 32, a Q4_0 / Q8_0 row) rides in the last
    # K = 256 over 8 devices: one shard holds it, the others are empty (they are skipped by the MUL_MAT)
    r, lo, hi, _ = ranges(1, 0, 8, None, 256)
    assert sum(1 for l, h in zip(lo, hi) if h > l) == 1


def test_bad_arguments_are_refused(ranges):
    assert ranges(0, 8, 8, None, 4096)[0] == -1                      # main device outside the node
    assert ranges(0, 0, 0, None, 4096)[0] == -1
    assert ranges(1, 0, 17, None, 4096)[0] == -1
    assert ranges(1, 0, 8, None, -1)[0] == -1


def test_k_ranges_agree_with_the_python_side_of_the_k_split(ranges):
    """ggml_amd/shard.py: k_range (what bench.py's ksplit_allreduce leg and the gloo tests shard by) and the plug-in agree on equal splits of whole superblocks"""
    import sys
    sys.path.insert(0, ROOT)
    from ggml_amd import shard
    for K in (8192, 4096, 14336):
        for world in (2, 4, 8):
            if (K // 256) % world:
                continue
            r, lo, hi, _ = ranges(1, 0, world, None, K)
            for rank in range(world):
                assert tuple(shard.k_range(K, rank, world)) == (lo[rank], hi[rank])
