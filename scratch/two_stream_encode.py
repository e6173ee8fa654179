"""Same k_mm8 encode launched back to back on 1 / 2 streams: separates tail overlap from phase effects."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
n, t, B, use_omega = bench.WORKLOADS["cfg3"]
d = t + 1; C = (B + d - 1) // d
ctx = Context.get(bench.BLS, 0)
shares0 = ctx.empty(B); shares0.random_(0, 1 << 61)
z, zc = list(range(d)), list(range(d, d + t))
for ns in (1, 2):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    ops, outs = [], []
    for s in streams:
        with torch.cuda.stream(s):
            ops.append(BatchOpen(bench.BLS, n, t, z=z, zc=zc, max_shares=B, device=0)); outs.append(ctx.empty(n * C))
    torch.cuda.synchronize()
    def step(i):
        k = i % ns
        with torch.cuda.stream(streams[k]):
            ops[k].r1_encode(shares0, out=outs[k])
    for i in range(20): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{ns} stream(s): {dt / 400 * 1e6:.1f} us per encode")
