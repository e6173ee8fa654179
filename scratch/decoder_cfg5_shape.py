"""DeviceIncrementalDecoder at config 5's shape (n = 256, t = 85, one GPU's shard: 6097 chunks, omega points): fault-free, 10 and 85 liars arriving first."""
import sys, time
import torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t = 256, 85
d = t + 1
C = 6097
ctx = Context.get(P)
gen = torch.Generator(device='cuda'); gen.manual_seed(3)
def rand(count):
    v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device='cuda', generator=gen); v[:, 3] &= (1 << 61) - 1; return v
coef = rand(C * d)
enc = BatchOpen(P, n, t, use_omega_powers=True, max_shares=C * d)
cols = enc.r1_encode(coef).view(n, C, 4).clone()
for liars in (0, 10, 85):
    data = cols.clone()
    for i in range(liars):
        data[i] = rand(C)
    order = list(range(n))
    for rep in range(3):
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, use_omega_powers=True, columns=data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        used = 0
        for idx in order:
            dec.add(idx); used += 1
            if dec.done(): break
        res, errs = dec.get_results()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = torch.equal(res.reshape(-1, 4), coef)
    print(f"n=256 t=85, {C * d} shares, {liars} liars: {dt*1e3:.2f} ms = {C*d/dt/1e6:.1f} M shares/s, {used} columns, errors {len(errs)}, probes {dec.probes}, quick {dec.quick_launches}, in-radius {dec.radius_verdicts}, exact {ok}", flush=True)
