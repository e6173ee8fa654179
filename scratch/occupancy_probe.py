"""R1-encode time vs number of 64-chunk groups: a jump after G = 256*k groups reveals how many
workgroups are co-resident per CU."""
import sys
import torch
sys.path.insert(0, '.')
import bench
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
BLS = bench.BLS
ctx = Context.get(BLS, 0)
n, t = 64, 21
d = t + 1
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
for G in (128, 256, 257, 384, 512, 513, 640, 768, 769, 900, 1024, 1025, 1280, 1536):
    C = 64 * G; B = C * d
    shares = bench.rand_elements(torch, B, gen)
    op = BatchOpen(BLS, n, t, z=list(range(d)), zc=list(range(d, d + t)), max_shares=B)
    out = ctx.empty(n * C)
    for _ in range(3): op.r1_encode(shares, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20; e0.record()
    for _ in range(K): op.r1_encode(shares, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"groups={G:5d}: r1_encode {e0.elapsed_time(e1)/K*1e3:7.1f} us", flush=True)
    del op, shares, out
