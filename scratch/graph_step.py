"""Does capturing one open into a HIP graph (torch.cuda.CUDAGraph) remove the inter-kernel gaps?"""
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
shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs(torch, ctx, n, t, B, use_omega, seed=1000)
order = np.random.Generator(np.random.PCG64(2024)).permutation(n).tolist()
z, zc = order[:d], order[d : d + t]
op = BatchOpen(bench.BLS, n, t, z=z, zc=zc, max_shares=B, device=0)
a, b, c = ctx.empty(n * C), ctx.empty(C), ctx.empty(B)
def step():
    op.r1_encode(shares0, out=a); op.r1_decode(r1_cols, B, out=b); op.r2_decode(r2_cols, B, out=c)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
torch.cuda.synchronize()
for _ in range(10): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert op.ok() and torch.equal(c, secrets)
print(f"graph replay: {dt / 200 * 1e3:.4f} ms per open  ({B * 200 / dt / 1e9:.3f} G shares/s)")
with torch.cuda.stream(s):
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"plain launches: {dt / 200 * 1e3:.4f} ms per open  ({B * 200 / dt / 1e9:.3f} G shares/s)")
