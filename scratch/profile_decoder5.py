"""Per-arrival wall clock of the 5-liar open (garbage everywhere, liars first) on the device decoder, config 3's shape."""
import sys, time
import torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.device import DeviceIncrementalDecoder
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t = 64, 21
d = t + 1
B = 1 << 20
C = (B + d - 1) // d
ctx = Context.get(P); lib = ctx.lib
gen = torch.Generator(device='cuda'); gen.manual_seed(3)
def rand(count):
    v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device='cuda', generator=gen); v[:, 3] &= (1 << 61) - 1; return v
coef = rand(C * d)
xh = ctx.host_elems(list(range(1, n + 1)))
cols = ctx.empty(n * C)
ctx.check(lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(xh), n, ctx.ptr(coef), C, d, ctx.ptr(cols), ctx.stream()), "enc")
cols = cols.view(C, n, 4).transpose(0, 1).contiguous()
liars = int(sys.argv[1]) if len(sys.argv) > 1 else 5
data = cols.clone()
for i in range(liars):
    data[i] = rand(C)
for rep in range(3):
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=C)
    torch.cuda.synchronize()
    marks = []
    t0 = time.perf_counter()
    for k in range(n):
        ta = time.perf_counter()
        dec.add(k, data[k])
        torch.cuda.synchronize()
        marks.append((time.perf_counter() - ta) * 1e6)
        if dec.done():
            break
    tot = time.perf_counter() - t0
    big = [(i, round(m)) for i, m in enumerate(marks) if m > 60]
    print(f"rep {rep}: total {tot*1e3:.2f} ms (with a sync per arrival), columns {len(marks)}, cheap adds avg {sum(m for m in marks if m <= 60)/max(1,sum(1 for m in marks if m <= 60)):.0f} us; costly: {big}; quick {dec.quick_launches} probes {dec.probes} radius {dec.radius_verdicts}", flush=True)
