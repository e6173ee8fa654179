"""
Chunk-axis sharding of one party's share vector over the GPUs of a node
(SURVEY.md 8e).  Every (t+1)-share chunk is independent, so rank r simply opens
chunks [lo_r, hi_r); there is no exchange inside the path.  An optional all-gather
assembles the opened values when one consumer needs them all.
"""


def shard_bounds(num_shares, d, world_size, rank):
    """Contiguous, chunk-aligned slice of `num_shares` shares for `rank`.
    Returns (first_share, end_share).  Chunk c = shares [c*d, (c+1)*d)."""
    chunks = (num_shares + d - 1) // d
    base, extra = divmod(chunks, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return min(lo * d, num_shares), min(hi * d, num_shares)


def all_gather_opened(local, num_shares, d, group=None):
    """Gather the per-rank opened slices (int64 tensors of shape (len, 4)) into the full
    (num_shares, 4) tensor on every rank.  Slices differ by at most one chunk, so the
    gather pads to the longest slice."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_bounds(num_shares, d, world, r) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    padded = torch.zeros((longest, 4), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)])[:num_shares]
