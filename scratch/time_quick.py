"""hb_quick_interp_check / probe kernels: back-to-back launches vs launches separated by idle gaps (clock / wake-up effects)."""
import sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t = 64, 21
d = t + 1
C = 47663
ctx = Context.get(P); lib = ctx.lib
gen = torch.Generator(device='cuda'); gen.manual_seed(3)
cols = torch.randint(-(1 << 63), (1 << 63) - 1, (n * C, 4), dtype=torch.int64, device='cuda', generator=gen); cols[:, 3] &= (1 << 61) - 1
xh = ctx.host_elems(list(range(1, n + 1)))
out = ctx.empty(C * d)
status = torch.tensor([0, (1 << 31) - 1], dtype=torch.int32, device='cuda')
z = np.arange(d, dtype=np.int32); zc = np.arange(d, d + t, dtype=np.int32)
def quick():
    ctx.check(lib.hb_quick_interp_check(ctx.h, np_ptr(xh), n, np_ptr(z), d, np_ptr(zc), t, ctx.ptr(cols), C, 0, ctx.ptr(out), ctx.ptr(status), ctx.stream()), "q")
quick(); torch.cuda.synchronize()
for gap in (0.0, 0.0005, 0.003):
    ts = []
    for _ in range(30):
        if gap:
            time.sleep(gap)
        t0 = time.perf_counter(); quick(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    print(f"gap {gap*1e3:.1f} ms: enqueue {np.median([a for a, _ in ts])*1e6:.0f} us, enqueue + complete {np.median([b for _, b in ts])*1e6:.0f} us")
t0 = time.perf_counter()
for _ in range(50):
    quick()
torch.cuda.synchronize()
print(f"50 back to back: {(time.perf_counter() - t0) / 50 * 1e6:.0f} us each")
