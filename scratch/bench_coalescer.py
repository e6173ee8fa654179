"""Mpc.open_share_array-style workload: K concurrent small opens per party, one by one vs coalesced (OpenCoalescer).
n parties in one process on one GPU (the transport is in-process queues): launch count and wall time per step."""
import asyncio, random, sys, time
import torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device_reconstruction import batch_reconstruct_device
from honeybadgermpc_amd.open_coalescer import OpenCoalescer
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t, K, size = 4, 1, 64, 256
ctx = Context.get(P)
rnd = random.Random(1)
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
def rand(count):
    v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device='cuda', generator=gen); v[:, 3] &= (1 << 61) - 1; return v
# shares of K arrays for every party: random values are fine for timing (the opens of honest parties agree on garbage-free data only
# if consistent, so build consistent ones: degree-t sharings evaluated with the dealer kernel)
from honeybadgermpc_amd.offline import ShareDealer
dealer = ShareDealer(P, n, t)
arrays = []
for k in range(K):
    shares, _ = dealer.deal_secrets(rand(size), generator=gen)
    arrays.append(shares.view(n, size, 4))
class Net:
    def __init__(self): self.q = [dict() for _ in range(n)]
    def qq(self, p, tag): return self.q[p].setdefault(tag, asyncio.Queue())
    def factory(self, i):
        def f(tag):
            def send(dest, msg): self.qq(dest, tag).put_nowait((i, msg))
            return send, self.qq(i, tag).get
        return f
async def one_by_one():
    net = Net()
    async def party(i):
        outs = []
        for k in range(K):
            send, recv = net.factory(i)(("open", k))
            outs.append(await batch_reconstruct_device(arrays[k][i].contiguous(), P, t, n, i, send, recv))
        return outs
    return await asyncio.gather(*[party(i) for i in range(n)])
async def coalesced():
    net = Net()
    async def party(i):
        co = OpenCoalescer(P, n, t, i, net.factory(i))
        hs = [co.open_share_array(arrays[k][i].contiguous()) for k in range(K)]
        return [await h for h in hs]
    return await asyncio.gather(*[party(i) for i in range(n)])
for name, fn in (("one batch_reconstruct per array", one_by_one), ("OpenCoalescer (one reconstruction)", coalesced)):
    asyncio.run(fn())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = asyncio.run(fn())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {K} opens x {size} shares, n={n} parties in one process: {dt*1e3:.1f} ms per step = {K*size*n/dt/1e6:.2f} M share-opens/s, {dt/K*1e3:.2f} ms per open", flush=True)
a = asyncio.run(one_by_one()); b = asyncio.run(coalesced())
print("results identical:", all(torch.equal(x, y) for pa, pb in zip(a, b) for x, y in zip(pa, pb)))
