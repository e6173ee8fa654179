"""N > 1 path on CPU: world-size-2 gloo.  Each rank opens its chunk-aligned slice (here
with the CPU oracle standing in for the per-rank device open) and the gathered result must
equal the unsharded open."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import BLS
from honeybadgermpc_amd.sharding import shard_bounds


def test_shard_bounds_cover_exactly():
    for b, d, w in [(1 << 20, 22, 8), (100, 6, 3), (5, 2, 4), (0, 3, 2), (7, 7, 2), (1 << 22, 86, 8)]:
        prev = 0
        for r in range(w):
            lo, hi = shard_bounds(b, d, w, r)
            assert lo == prev and lo <= hi and (lo % d == 0 or lo == b)
            prev = hi
        assert prev == b
    lo, hi = shard_bounds(1 << 20, 22, 8, 3)
    assert abs((hi - lo) - (1 << 20) / 8) <= 22


class OracleOpener:
    """CPU stand-in with device.BatchOpen's interface (tensors of 4 x u64 limbs viewed as int64), computing with the oracle:
    lets the gloo test run honeybadgermpc_amd.sharding.ShardedOpen -- the code bench.py --workload cfg5 runs -- without a GPU."""

    def __init__(self, p, n, t, max_shares=0, z=None, zc=None, use_omega_powers=False, degree=None, device=None):
        import oracle

        assert not use_omega_powers
        self.o, self.p, self.n, self.t = oracle, p, n, t
        self.d = (t if degree is None else degree) + 1
        self.x = list(range(1, n + 1))
        self.z = list(range(self.d)) if z is None else list(z)
        self.zc = [i for i in range(n) if i not in self.z][:t] if zc is None else list(zc)
        self.fine = True

    def _ints(self, tensor):
        import numpy as np

        return self.o._ints(tensor.numpy().view(np.uint64))

    def _tensor(self, values):
        import numpy as np

        return torch.from_numpy(self.o._limbs(values, self.p).view(np.int64).copy())

    def r1_encode(self, shares, out=None):
        vals = self._ints(shares)
        c = (len(vals) + self.d - 1) // self.d
        rows = [(vals[i * self.d : (i + 1) * self.d] + [0] * self.d)[: self.d] for i in range(c)]
        enc = self.o.vandermonde_batch_evaluate(self.x, rows, self.p)                # [c][n]
        return self._tensor([enc[k][j] for j in range(self.n) for k in range(c)])    # party-major [n][c]

    def _decode(self, cols, b):
        vals = self._ints(cols)
        c = (b + self.d - 1) // self.d
        col = lambda j: vals[j * c : (j + 1) * c]  # noqa: E731
        data = [[col(j)[k] for j in self.z] for k in range(c)]
        coef = self.o.vandermonde_batch_interpolate([self.x[j] for j in self.z], data, self.p)
        enc = self.o.vandermonde_batch_evaluate(self.x, coef, self.p)
        for j in self.zc:
            if any(enc[k][j] != col(j)[k] for k in range(c)):
                self.fine = False
        return coef

    def r1_decode(self, r1_cols, b, out=None):
        return self._tensor([row[0] for row in self._decode(r1_cols, b)])

    def r2_decode(self, r2_cols, b, out=None):
        return self._tensor([v for row in self._decode(r2_cols, b) for v in row][:b])

    def ok(self):
        fine, self.fine = self.fine, True
        return fine


def _worker_sharded_open(rank, world, port, b, n, t, shares, r1_cols, r2_cols, expect, mode, ret):
    """rank's slice through sharding.ShardedOpen (stand-in opener), then the data-path gather"""
    import numpy as np

    import oracle
    from honeybadgermpc_amd.sharding import ShardedOpen

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = t + 1
    z, zc = list(range(1, d + 1)), list(range(d + 1, min(n, d + 1 + t)))
    so = ShardedOpen(BLS, n, t, b, gather_mode=mode, make_opener=OracleOpener, z=z, zc=zc)
    to_t = lambda vals: torch.from_numpy(oracle._limbs(vals, BLS).view(np.int64).copy())  # noqa: E731
    clo, chi = so.chunk_lo, so.chunk_lo + so.chunks
    mine = to_t(shares[so.lo : so.hi])
    r1 = so.r1_encode(mine)
    assert r1.shape[0] == n * so.chunks
    msg = so.r1_decode(to_t([v for col in r1_cols for v in col[clo:chi]]))
    res = so.r2_decode(to_t([v for col in r2_cols for v in col[clo:chi]]))
    ok = so.ok() and msg.shape[0] == so.chunks and res.shape[0] == so.hi - so.lo
    full = so.gather(res)
    got = oracle._ints(full.numpy().view(np.uint64))
    # a corrupted validated column on ONE rank is reported by that rank only
    bad = [list(col) for col in r2_cols]
    if rank == 1 and so.chunks:
        bad[zc[0]][clo] = (bad[zc[0]][clo] + 1) % BLS
    so.r2_decode(to_t([v for col in bad for v in col[clo:chi]]))
    flagged = not so.ok()
    ret[rank] = bool(ok and got == expect and flagged == (rank == 1 and so.chunks > 0))
    dist.destroy_process_group()


def _worker(rank, world, port, b, d, n, t, shares, r1_cols, r2_cols, expect, ret):
    import numpy as np

    import oracle
    from honeybadgermpc_amd.sharding import all_gather_opened

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(b, d, world, rank)
    c_all = (b + d - 1) // d
    clo, chi = lo // d, (hi + d - 1) // d
    x = list(range(1, n + 1))
    lim = lambda rows: oracle._limbs([v for r in rows for v in r], BLS)  # noqa: E731
    cols1 = [col[clo:chi] for col in r1_cols]
    cols2 = [col[clo:chi] for col in r2_cols]
    rc, _, _, res = oracle.batch_open_limbs(BLS, n, d, x, oracle._limbs(shares[lo:hi], BLS), lim(cols1), lim(cols2),
                                            list(range(d)), list(range(d, d + t)))
    assert rc == 0
    local = torch.from_numpy(res.view(np.int64).copy())
    full = all_gather_opened(local, b, d)
    got = oracle._ints(full.numpy().view(np.uint64))
    ret[rank] = got == expect and c_all == (b + d - 1) // d
    dist.destroy_process_group()


def test_two_rank_sharded_open_matches_unsharded():
    import oracle

    rnd = random.Random(4)
    n, t, b = 7, 2, 50
    d = t + 1
    c = (b + d - 1) // d
    x = list(range(1, n + 1))
    polys1 = [[rnd.randrange(BLS) for _ in range(d)] for _ in range(c)]
    polys2 = [[rnd.randrange(BLS) for _ in range(d)] for _ in range(c)]
    e1 = oracle.vandermonde_batch_evaluate(x, polys1, BLS)
    e2 = oracle.vandermonde_batch_evaluate(x, polys2, BLS)
    r1_cols = [[e1[k][j] for k in range(c)] for j in range(n)]
    r2_cols = [[e2[k][j] for k in range(c)] for j in range(n)]
    shares = [rnd.randrange(BLS) for _ in range(b)]
    expect = [v for row in polys2 for v in row][:b]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, b, d, n, t, shares, r1_cols, r2_cols, expect, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


@pytest.mark.parametrize("mode,world,b", [("direct", 2, 50), ("collective", 2, 50), ("direct", 3, 31), ("collective", 3, 31), ("direct", 2, 48)])
def test_sharded_open_class_on_gloo(mode, world, b):
    """the class bench.py --workload cfg5 and the GPU test drive, with a CPU stand-in opener: uneven slices, both gathers"""
    import oracle

    rnd = random.Random(9 + b)
    n, t = 7, 2
    d = t + 1
    c = (b + d - 1) // d
    x = list(range(1, n + 1))
    polys1 = [[rnd.randrange(BLS) for _ in range(d)] for _ in range(c)]
    polys2 = [[rnd.randrange(BLS) for _ in range(d)] for _ in range(c)]
    e1 = oracle.vandermonde_batch_evaluate(x, polys1, BLS)
    e2 = oracle.vandermonde_batch_evaluate(x, polys2, BLS)
    r1_cols = [[e1[k][j] for k in range(c)] for j in range(n)]
    r2_cols = [[e2[k][j] for k in range(c)] for j in range(n)]
    shares = [rnd.randrange(BLS) for _ in range(b)]
    expect = [v for row in polys2 for v in row][:b]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_sharded_open, args=(r, world, port, b, n, t, shares, r1_cols, r2_cols, expect, mode, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))


def _worker_subgroup(rank, world, port, b, d, mode, ret):
    """all_gather_opened inside a NON-default group (ranks 1..world-1): P2POp peers are global ranks, slices are indexed by
    the rank within the group"""
    from honeybadgermpc_amd.sharding import all_gather_opened

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    members = list(range(1, world))
    group = dist.new_group(ranks=members)
    good = True
    if rank in members:
        gr = dist.get_rank(group)
        gw = len(members)
        lo, hi = shard_bounds(b, d, gw, gr)
        whole = torch.arange(b * 4, dtype=torch.int64).reshape(b, 4) * 7 + 3
        full = all_gather_opened(whole[lo:hi].clone(), b, d, group=group, mode=mode)
        good = torch.equal(full, whole)
    ret[rank] = bool(good)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["direct", "collective"])
def test_gather_inside_a_subgroup(mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, 3, port, 31, 3, mode, ret)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(3))


# ---- world 8: the first hardware scaling run must not be the first time rank 7 exists -------------------
def test_config5_split_over_eight_ranks():
    """BASELINE config 5: 2^22 shares, d = 86 -> 48 771 chunks; over 8 ranks that is uneven (48 771 = 8 * 6096 + 3)"""
    b, d, w = 1 << 22, 86, 8
    bounds = [shard_bounds(b, d, w, r) for r in range(w)]
    chunks = [(hi - lo + d - 1) // d for lo, hi in bounds]
    assert chunks == [6097, 6097, 6097, 6096, 6096, 6096, 6096, 6096] and sum(chunks) == 48771
    assert bounds[0] == (0, 6097 * 86) and bounds[-1][1] == b
    assert bounds[-1][1] - bounds[-1][0] == b - (3 * 6097 + 4 * 6096) * 86          # the last rank's slice ends inside its last chunk
    for (lo, hi), (lo2, _) in zip(bounds, bounds[1:]):
        assert hi == lo2 and lo % d == 0


def _worker_gather_real_split(rank, world, port, b, d, mode, ret):
    """all_gather_opened with config 5's real slice lengths; element (i, j) = 4 i + j so every position is checkable"""
    from honeybadgermpc_amd.sharding import all_gather_opened

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(b, d, world, rank)
    local = (torch.arange(lo, hi, dtype=torch.int64).unsqueeze(1) * 4 + torch.arange(4, dtype=torch.int64).unsqueeze(0)).contiguous()
    full = all_gather_opened(local, b, d, mode=mode)
    # spot checks instead of a second 134 MB tensor: the ends of every rank's slice and a stride through the middle
    idx = torch.tensor(sorted({0, b - 1} | {p for r in range(world) for p in shard_bounds(b, d, world, r) if 0 <= p < b}
                              | {p - 1 for r in range(world) for p in shard_bounds(b, d, world, r) if p > 0} | set(range(0, b, 100003))))
    good = bool(torch.equal(full[idx], idx.unsqueeze(1) * 4 + torch.arange(4).unsqueeze(0))) and tuple(full.shape) == (b, 4)
    ret[rank] = good
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,b", [("direct", 1 << 22), ("collective", 1 << 22)])
def test_gather_with_config5_real_split_world_8(mode, b):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_gather_real_split, args=(r, 8, port, b, 86, mode, ret)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(8))


@pytest.mark.parametrize("mode", ["direct", "collective"])
def test_sharded_open_class_world_8_uneven(mode):
    """sharding.ShardedOpen on eight ranks with a chunk count that splits as config 5's does (8 k + 3 chunks), stand-in opener"""
    import oracle

    b = 56                                         # d = 3 -> 19 chunks = 8 * 2 + 3, the last one short
    rnd = random.Random(8)
    n, t = 7, 2
    d = t + 1
    c = (b + d - 1) // d
    x = list(range(1, n + 1))
    polys1 = [[rnd.randrange(BLS) for _ in range(d)] for _ in range(c)]
    polys2 = [[rnd.randrange(BLS) for _ in range(d)] for _ in range(c)]
    e1 = oracle.vandermonde_batch_evaluate(x, polys1, BLS)
    e2 = oracle.vandermonde_batch_evaluate(x, polys2, BLS)
    r1_cols = [[e1[k][j] for k in range(c)] for j in range(n)]
    r2_cols = [[e2[k][j] for k in range(c)] for j in range(n)]
    shares = [rnd.randrange(BLS) for _ in range(b)]
    expect = [v for row in polys2 for v in row][:b]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_sharded_open, args=(r, 8, port, b, n, t, shares, r1_cols, r2_cols, expect, mode, ret)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(8))
