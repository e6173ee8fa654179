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
samples = []
orig_decide = device._QuickDec.decide
def decide(self, zc, cols_, c, out):
    t0 = time.perf_counter(); r = orig_decide(self, zc, cols_, c, out); dt_ = time.perf_counter() - t0; acc["decide"] = acc.get("decide", 0) + dt_; samples.append(dt_); return r
device._QuickDec.decide = decide
orig_empty = type(ctx).empty
def empty(self, count):
    t0 = time.perf_counter(); r = orig_empty(self, count); acc["empty"] = acc.get("empty", 0) + time.perf_counter() - t0; return r
type(ctx).empty = empty
for want in ("constant", "all", "constant", "all", "constant"):
    acc.clear(); samples.clear()
    for rep in range(100):
        order = rng.permutation(n).tolist()
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, columns=cols, want=want)
        for idx in order:
            dec.add(idx)
            if dec.done(): break
        res = dec.get_results()[0]
    ss = sorted(samples)
    print(want, {k: round(v / 100 * 1e6, 1) for k, v in acc.items()}, "decide min/med/p90/max us:", [round(x * 1e6) for x in (ss[0], ss[50], ss[90], ss[-1])], flush=True)
