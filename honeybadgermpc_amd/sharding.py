"""
Chunk-axis sharding of one party's batch open over the GPUs of a node (BASELINE config 5:
"batch_reconstruction n=256 t=85, 2^22 shares sharded across 8 MI355X via RCCL/xGMI").

Every (t+1)-share chunk is independent -- the only shared data are the O(n d) tables, replicated per GPU -- so rank r
opens chunks [lo_r, hi_r) of the share vector (reference batch_reconstruction.py:158-227 on its slice) and nothing is
exchanged inside the path.  When one consumer needs every opened value, the slices are all-gathered: that is the
data-path collective of the strong-scaling mode, and it is timed.

    direct      every rank posts its slice to each of its peers at once (batch_isend_irecv): on the xGMI full mesh the
                seven transfers of a rank run on seven links concurrently (~0.11 ms for config 5 at 8 GPUs)
    collective  torch.distributed.all_gather_into_tensor (RCCL picks ring / tree), slices padded to the longest

`ShardedOpen` is what bench.py --workload cfg5 and the tests drive.  The opener is pluggable (default: device.BatchOpen)
so that the world-size-2 gloo test on CPU runs this very code with a stand-in opener.
"""


def shard_bounds(num_shares, d, world_size, rank):
    """Contiguous, chunk-aligned slice of `num_shares` shares for `rank`.
    Returns (first_share, end_share).  Chunk c = shares [c*d, (c+1)*d)."""
    chunks = (num_shares + d - 1) // d
    base, extra = divmod(chunks, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return min(lo * d, num_shares), min(hi * d, num_shares)


def all_gather_opened(local, num_shares, d, group=None, mode="collective", out=None):
    """Gather the per-rank opened slices (int64 tensors of shape (len, limbs)) into the full (num_shares, limbs)
    tensor on every rank.  Slices differ by at most one chunk."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(num_shares, d, world, r) for r in range(world)]
    limbs = local.shape[1]
    if out is None:
        out = torch.empty((num_shares, limbs), dtype=local.dtype, device=local.device)
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} opened shares, its slice has {hi - lo}")
    if mode == "direct":
        out[lo:hi] = local
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            plo, phi = sizes[peer]
            # P2POp addresses peers by GLOBAL rank; `peer` counts within `group`
            gpeer = peer if group is None else dist.get_global_rank(group, peer)
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, local, gpeer, group))
            if phi > plo:
                ops.append(dist.P2POp(dist.irecv, out[plo:phi], gpeer, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    if mode != "collective":
        raise ValueError("mode must be 'collective' or 'direct'")
    longest = max(h - l for l, h in sizes)
    if all(h - l == longest for l, h in sizes) and longest * world == num_shares:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((longest, limbs), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = torch.empty((world * longest, limbs), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(parts, padded, group=group)
    for r, (l, h) in enumerate(sizes):
        out[l:h] = parts[r * longest : r * longest + (h - l)]
    return out


class ShardedOpen:
    """This rank's part of one party's batch open of `num_shares` shares split over the process group.

        so = ShardedOpen(p, n, t, num_shares, z=z, zc=zc, use_omega_powers=...)
        r1 = so.r1_encode(my_shares)             # my_shares: this rank's slice [so.lo, so.hi)
        msg = so.r1_decode(r1_cols)              # r1_cols: [n][so.chunks] party-major, this rank's chunk range
        res = so.r2_decode(r2_cols)              # (so.hi - so.lo, limbs)
        full = so.gather(res)                    # every rank: (num_shares, limbs); asserts so.ok() first is the caller's call

    make_opener(p, n, t, max_shares=..., **kw) must return an object with BatchOpen's r1_encode / r1_decode /
    r2_decode / ok (device.BatchOpen by default)."""

    def __init__(self, modulus, n, t, num_shares, group=None, gather_mode="direct", make_opener=None, **kw):
        import torch.distributed as dist

        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_shares = int(num_shares)
        degree = kw.get("degree")
        self.d = (t if degree is None else degree) + 1
        self.lo, self.hi = shard_bounds(self.num_shares, self.d, self.world, self.rank)
        self.chunk_lo = self.lo // self.d
        self.chunks = (self.hi - self.lo + self.d - 1) // self.d
        self.gather_mode = gather_mode
        if make_opener is None:
            from .device import BatchOpen as make_opener
        self.op = make_opener(modulus, n, t, max_shares=max(self.hi - self.lo, 1), **kw)

    @property
    def local_shares(self):
        return self.hi - self.lo

    def r1_encode(self, shares, out=None):
        return self.op.r1_encode(shares, out=out)

    def r1_decode(self, r1_cols, out=None):
        return self.op.r1_decode(r1_cols, self.local_shares, out=out)

    def r2_decode(self, r2_cols, out=None):
        return self.op.r2_decode(r2_cols, self.local_shares, out=out)

    def ok(self):
        return self.op.ok()

    def gather(self, local_result, out=None, through_collective=False):
        """through_collective: a group of ONE rank still goes through the transport (RCCL's all_gather_into_tensor, or batch_isend_irecv with no
        peers): the N = 1 point of a scaling run then exercises the same calls as N = 2, 4, 8 (bench.py)"""
        if self.world == 1 and not through_collective:
            return local_result
        return all_gather_opened(local_result, self.num_shares, self.d, self.group, self.gather_mode, out=out)
