"""cProfile of the add() that meets the first disagreement (the optimistic launch fails: candidates, the probe's start) and of the last one
(the verdict's launch), at config 3's or config 5's shard shape with t liars first or spread:  python scratch/profile_first_disagreement.py [n t [spread]]"""
import cProfile, pstats, sys, time, io
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.device import DeviceIncrementalDecoder
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t = (64, 21) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
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
if len(sys.argv) > 3 and sys.argv[3] == 'spread':  # (argv[4]: first | mid | last -- which add() is profiled)
    honest = list(range(t, n)); step = len(honest) // (t + 1); order = []
    for i in range(t):
        order += honest[i * step:(i + 1) * step] + [i]
    order += honest[t * step:]
import gc
gc.collect(); gc.freeze()
need = d + t
prof = cProfile.Profile()
for rep in range(8):
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, columns=data, use_omega_powers=omega)
    torch.cuda.synchronize()
    which = sys.argv[4] if len(sys.argv) > 4 else "first"
    for k_, idx in enumerate(order):
        if rep >= 3 and ((which == "first" and k_ == need - 1) or (which == "last" and k_ == n - 1) or (which == "mid" and k_ == need + (n - need) // 2)):
            prof.enable(); dec.add(idx); prof.disable()
        else:
            dec.add(idx)
        if dec.done(): break
    torch.cuda.synchronize()
st = pstats.Stats(prof)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][3])[:32]
reps = 5
print(f"per add() that meets the first disagreement, us (profiled: {reps} adds)   cumulative   own   calls")
for (fn, line, name), (cc, nc, tt, ct, callers) in rows:
    print(f"{ct / reps * 1e6:9.1f} {tt / reps * 1e6:8.1f} {nc / reps:6.1f}  {fn.split('/')[-1]}:{line}({name})")
