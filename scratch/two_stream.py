"""Does overlapping consecutive opens on two streams (two plans, own buffers) raise the throughput?"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen  # noqa: E402

n, t, B, use_omega = bench.WORKLOADS["cfg3"]
d = t + 1
C = (B + d - 1) // d
ctx = Context.get(bench.BLS, 0)
shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs(torch, ctx, n, t, B, use_omega, seed=1000)
order = np.random.Generator(np.random.PCG64(2024)).permutation(n).tolist()
z, zc = order[:d], order[d : d + t]
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    ops, bufs = [], []
    for s in streams:
        with torch.cuda.stream(s):
            ops.append(BatchOpen(bench.BLS, n, t, z=z, zc=zc, max_shares=B, device=0))
            bufs.append((ctx.empty(n * C), ctx.empty(C), ctx.empty(B)))
    torch.cuda.synchronize()

    def step(i):
        k = i % ns
        with torch.cuda.stream(streams[k]):
            op, (a, b, c) = ops[k], bufs[k]
            op.r1_encode(shares0, out=a)
            op.r1_decode(r1_cols, B, out=b)
            op.r2_decode(r2_cols, B, out=c)

    for i in range(12):
        step(i)
    torch.cuda.synchronize()
    steps = 240
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    oks = [op.ok() for op in ops]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(oks) and all(torch.equal(b[2], secrets) for b in bufs)
    print(f"{ns} stream(s): {B * steps / dt / 1e9:.3f} G shares/s  ({dt / steps * 1e3:.4f} ms per open)")
