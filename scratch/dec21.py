"""The 21-liar open at config 3's shape alone (for kernel traces): liars send garbage everywhere and arrive first
(third argument `spread`: one liar after every few honest senders instead -- no candidate stands, the probe decides)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.device import DeviceIncrementalDecoder
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t = (64, 21) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
omega = n > 100
d = t + 1
B = (1 << 20) if n == 64 else (1 << 22) // 8
C = (B + d - 1) // d
ctx = Context.get(P); lib = ctx.lib
gen = torch.Generator(device='cuda'); gen.manual_seed(3)
def rand(count):
    v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device='cuda', generator=gen); v[:, 3] &= (1 << 61) - 1; return v
coef = rand(C * d)
if omega:
    from honeybadgermpc_amd.device import BatchOpen
    cols = BatchOpen(P, n, t, use_omega_powers=True, max_shares=C * d).r1_encode(coef).view(n, C, 4).clone()
else:
    xh = ctx.host_elems(list(range(1, n + 1)))
    cols = ctx.empty(n * C)
    ctx.check(lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(xh), n, ctx.ptr(coef), C, d, ctx.ptr(cols), ctx.stream()), "enc")
    cols = cols.view(C, n, 4).transpose(0, 1).contiguous()
data = cols.clone()
for i in range(t):
    data[i] = rand(C)
order = list(range(n))
if len(sys.argv) > 3 and sys.argv[3] == 'spread':
    honest = list(range(t, n)); step = len(honest) // (t + 1); order = []
    for i in range(t):
        order += honest[i * step:(i + 1) * step] + [i]
    order += honest[t * step:]
import gc
if len(sys.argv) > 4 and sys.argv[4] == "freeze":
    gc.collect(); gc.freeze()
for rep in range(6):
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, columns=data, use_omega_powers=omega)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marks = []
    for idx in order:
        t1 = time.perf_counter(); dec.add(idx); marks.append(time.perf_counter() - t1)
        if dec.done(): break
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{dt*1e3:.2f} ms ({B / dt / 1e6:.0f} M shares/s; probes {dec.probes}, quick {dec.quick_launches}, in-radius {dec.radius_verdicts}); add() times us:", [round(m * 1e6) for m in marks])
