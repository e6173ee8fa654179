"""hb_symbols_fetch alone, and one add() of the waiting-candidate state by its parts (cProfile)."""
import sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
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
out = np.empty((1, 4), dtype=np.int64)
ia = np.asarray([5], dtype=np.int32)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(1000):
        ctx.lib.hb_symbols_fetch(ctx.h, ctx.ptr(cols), n, C, 7, np_ptr(ia), 1, np_ptr(out), ctx.stream())
    dt = time.perf_counter() - t0
    print(f"hb_symbols_fetch: {dt * 1e3:.2f} us per call (incl. ctx.ptr / ctx.stream / np_ptr)")
st = ctx.stream(); pc = ctx.ptr(cols); pi = np_ptr(ia); po = np_ptr(out)
t0 = time.perf_counter()
for i in range(1000):
    ctx.lib.hb_symbols_fetch(ctx.h, pc, n, C, 7, pi, 1, po, st)
print(f"hb_symbols_fetch, arguments prepared: {(time.perf_counter() - t0) * 1e3:.2f} us per call")
data = cols.clone()
for i in range(t):
    data[i] = rand(C)
for rep in range(3):
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, use_omega_powers=True, columns=data)
    for idx in range(200):
        dec.add(idx)
    torch.cuda.synchronize()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for idx in range(200, 250):
        dec.add(idx)
    dt = time.perf_counter() - t0
    if rep == 2:
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
    print(f"add() while a candidate waits: {dt / 50 * 1e6:.1f} us")
