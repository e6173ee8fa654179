import sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd import ntl
from honeybadgermpc_amd._capi import Context, np_ptr
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, d, c = 64, 22, 47663
ctx = Context.get(P)
x = list(range(1, n + 1))
a = ctx.empty(c * d); a.random_(0, 1 << 62)
arr = a.view(c, d, 4)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ntl.vandermonde_batch_evaluate(x, arr, P)
    torch.cuda.synchronize(); print("ntl evaluate (device tensor):", (time.perf_counter() - t0) * 1e3, "ms")
xh = ctx.host_elems(x)
dout = ctx.empty(c * n)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = ctx.lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(xh), n, ctx.ptr(a), c, d, ctx.ptr(dout), ctx.stream())
    torch.cuda.synchronize(); print("C call:", (time.perf_counter() - t0) * 1e3, "ms", rc)
t0 = time.perf_counter(); xh2 = ctx.host_elems(x); print("host_elems:", (time.perf_counter() - t0) * 1e3, "ms")
rows = out[:, :d].contiguous()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec = ntl.vandermonde_batch_interpolate(x[:d], rows, P)
    torch.cuda.synchronize(); print("ntl interpolate (device tensor):", (time.perf_counter() - t0) * 1e3, "ms")
