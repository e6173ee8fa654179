import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.device import DeviceIncrementalDecoder
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t = 64, 21
d = t + 1
for B in (1 << 20, 1 << 14):
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
    for want in ("constant", "all", "constant", "constant"):
        order = rng.permutation(n).tolist()
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, columns=cols, want=want)
        for idx in order:
            dec.add(idx)
            if dec.done(): break
        res = dec.get_results()[0]
        ok = torch.equal(res[:, 0, :], coef.view(C, d, 4)[:, 0, :])
        print(B, want, "shape", tuple(res.shape), "optimistic", dec._optimistic, "quick launches", dec.quick_launches, "ok", ok, flush=True)
