"""Where the host time of a first-sight decode goes (config 3 shape): constructor, add() calls, the two C calls."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd import device
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
rng = np.random.Generator(np.random.PCG64(5))
acc = {}
def T(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
for want in (sys.argv[1:] or ["all", "constant"]):
    acc.clear(); lasts = []
    reps = 200
    for rep in range(reps + 5):
        if rep == 5:
            acc.clear(); torch.cuda.synchronize(); t_all = time.perf_counter()
        order = rng.permutation(n).tolist()
        t0 = time.perf_counter()
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, columns=cols, want=want)
        T("ctor", t0)
        for k, idx in enumerate(order):
            t0 = time.perf_counter()
            dec.add(idx)
            T("add_d" if k == d - 1 else ("add_last" if k == d + t - 1 else "add_other"), t0)
            if k == d + t - 1:
                lasts.append(time.perf_counter() - t0)
            if dec.done():
                break
        t0 = time.perf_counter()
        res = dec.get_results()[0]
        T("results", t0)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all) / reps
    ls = [round(v * 1e6) for v in lasts[5:]]
    print(want, f"total {tot*1e6:.1f} us per decode;", {k: round(v / reps * 1e6, 1) for k, v in acc.items()}, "add_last us: first 8", ls[:8], "sorted deciles", sorted(ls)[::20])
    if want == "all":
        assert torch.equal(res.reshape(-1, 4), coef)
