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
