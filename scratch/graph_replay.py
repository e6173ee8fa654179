"""One per-party open (3 launches) captured in a HIP graph and replayed, against the same launches issued one by one."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
P = bench.BLS
n, t, B = 64, 21, 1 << 20
d = t + 1; C = (B + d - 1) // d
ctx = Context.get(P)
shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs_light(torch, ctx, n, t, B, False, seed=5)
order = np.random.Generator(np.random.PCG64(2024)).permutation(n).tolist()
op = BatchOpen(P, n, t, z=order[:d], zc=order[d:d + t], max_shares=B)
op.set_fused_validate(True)
r1_out, r2_msg, result = ctx.empty(n * C), ctx.empty(C), ctx.empty(B)
def step():
    op.r1_encode(shares0, out=r1_out); op.r1_decode(r1_cols, B, out=r2_msg); op.r2_decode(r2_cols, B, out=result)
for _ in range(5): step()
assert op.ok() and torch.equal(result, secrets)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"launch by launch: {dt*1e6:.1f} us per open = {B/dt/1e9:.2f} G shares/s")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
assert op.ok() and torch.equal(result, secrets)
print(f"graph replay:     {dt*1e6:.1f} us per open = {B/dt/1e9:.2f} G shares/s")
