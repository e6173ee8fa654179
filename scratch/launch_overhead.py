"""Kernel time vs batch size for the R1 encode and the validating re-encode: separates the fixed
per-launch cost from the per-chunk cost."""
import sys, time
import torch
sys.path.insert(0, '.')
import bench
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
BLS = bench.BLS
ctx = Context.get(BLS, 0)
n, t = 64, 21
d = t + 1
for logb in (20, 19, 18, 17, 16, 14, 12, 10):
    B = 1 << logb
    shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs(torch, ctx, n, t, B, False, seed=1)
    C = (B + d - 1) // d
    op = BatchOpen(BLS, n, t, z=list(range(d)), zc=list(range(d, d + t)), max_shares=B)
    r1_out = ctx.empty(n * C); r2_msg = ctx.empty(C); res = ctx.empty(B)
    for _ in range(3):
        op.r1_encode(shares0, out=r1_out); op.r1_decode(r1_cols, B, out=r2_msg)
    torch.cuda.synchronize()
    K = 20
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tenc = tdec = 0.0
    for _ in range(K):
        e[0].record(); op.r1_encode(shares0, out=r1_out); e[1].record(); op.r1_decode(r1_cols, B, out=r2_msg); e[2].record()
        torch.cuda.synchronize()
        tenc += e[0].elapsed_time(e[1]); tdec += e[1].elapsed_time(e[2])
    assert op.ok()
    print(f"B=2^{logb} C={C}: r1_encode {tenc/K*1e3:8.1f} us   r1_decode(decode+validate) {tdec/K*1e3:8.1f} us", flush=True)
