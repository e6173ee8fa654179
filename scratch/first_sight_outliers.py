"""Per-decode wall time of many first-sight decodes (fresh arrival order each): are there periodic stalls?"""
import sys, time, gc
import numpy as np, torch
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
rng = np.random.Generator(np.random.PCG64(5))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
want = sys.argv[2] if len(sys.argv) > 2 else "all"
if len(sys.argv) > 3:
    gc.disable()
ts = []
for rep in range(reps):
    order = rng.permutation(n).tolist()
    t0 = time.perf_counter()
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, columns=cols, want=want)
    for idx in order:
        dec.add(idx)
        if dec.done():
            break
    res = dec.get_results()[0]
    ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
a = np.array(ts[5:]) * 1e6
print(f"{want}: {len(a)} decodes, mean {a.mean():.1f} us, median {np.median(a):.1f}, p99 {np.percentile(a, 99):.1f}, max {a.max():.0f}; gc {'off' if len(sys.argv) > 3 else 'on'}")
print("decodes over 1 ms (index, us):", [(i + 5, int(v)) for i, v in enumerate(a) if v > 1000])
