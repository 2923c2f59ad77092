"""Row-split (output-feature) sharding of a quantized weight matrix across the GPUs of one node — the partition of
the reference's split buffer type (src/ggml-cuda/ggml-cuda.cu:716-742: contiguous row ranges, boundaries rounded
to the GEMM row tile).  Pure host arithmetic + torch.distributed plumbing; no kernels here."""
import torch

ROW_TILE = 128     # m-tile of k_gemm_q / k_gemm_kq_pipe


def row_range(M, rank, world, align=ROW_TILE):
    """[lo, hi) of the rows owned by `rank`: equal split, boundaries rounded down to `align` (last rank takes the
    remainder), like get_row_rounding / the tensor_split loop of the reference."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    def bound(r):
        if r >= world:
            return M
        b = (M * r) // world
        return b - b % align
    return bound(rank), bound(rank + 1)


def all_row_ranges(M, world, align=ROW_TILE):
    return [row_range(M, r, world, align) for r in range(world)]


def gather_rows(y_local, M, world, group=None, align=ROW_TILE):
    """all-gather the per-rank output shards y_local (B, M_r) into the full (B, M) on every rank.
    One all_gather_into_tensor (one RCCL call over the fully connected xGMI mesh); ragged shards are padded."""
    import torch.distributed as dist
    B = y_local.shape[0]
    ranges = all_row_ranges(M, world, align)
    sizes = [hi - lo for lo, hi in ranges]
    if len(set(sizes)) == 1:
        buf = torch.empty((world * B, sizes[0]), dtype=y_local.dtype, device=y_local.device)   # rank-major concatenation
        dist.all_gather_into_tensor(buf, y_local.contiguous(), group=group)
        return buf.view(world, B, sizes[0]).permute(1, 0, 2).reshape(B, M)
    mx = max(sizes)                                   # ragged shards: pad to the widest, gather, cut
    pad = torch.zeros((B, mx), dtype=y_local.dtype, device=y_local.device)
    pad[:, :y_local.shape[1]] = y_local
    buf = torch.empty((world * B, mx), dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    buf = buf.view(world, B, mx)
    return torch.cat([buf[r, :, :sizes[r]] for r in range(world)], dim=1)


def k_range(K, rank, world, block=256):
    """[lo, hi) of the K (input-feature) range owned by `rank` in the K-split variant of a layer: whole `block`s (superblocks
    for K-quants), as evenly as they go — the first (K / block) % world ranks take one more.  Every rank then computes a full-size
    PARTIAL output over its K range and the partials are summed with one all-reduce."""
    if world <= 0 or not (0 <= rank < world) or K % block:
        raise ValueError("bad rank/world/K")
    nb = K // block
    base, rem = divmod(nb, world)
    lo = rank * base + min(rank, rem)
    return lo * block, (lo + base + (1 if rank < rem else 0)) * block


def k_shard_bytes(w_bytes, M, K, block, block_bytes, rank, world):
    """the K-shard of a block-quantized [M x K] matrix (numpy uint8, rows contiguous): blocks run along K inside each row, so a K
    range is the SAME byte range of every row — a strided slice, not a contiguous one (that is why the reference's set_tensor
    would have to scatter per-row sub-ranges)."""
    lo, hi = k_range(K, rank, world, block)
    rb = K // block * block_bytes
    return w_bytes.reshape(M, rb)[:, lo // block * block_bytes:hi // block * block_bytes].copy().reshape(-1), lo, hi
