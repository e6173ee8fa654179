"""Fixed cost of a k_matvec3 launch: same LDS footprint (d = 22 terms -> 50 KB per workgroup) and grid
(768 workgroups) as the config-3 encode, but only n_out outputs of work per workgroup."""
import ctypes, sys
import torch
sys.path.insert(0, '.')
import bench
from honeybadgermpc_amd._capi import Context, np_ptr
BLS = bench.BLS
ctx = Context.get(BLS, 0); lib = ctx.lib
d = 22
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
for G in (256, 768):
    C = 64 * G
    polys = bench.rand_elements(torch, C * d, gen)
    for n in (1, 4, 16, 64):
        x = ctx.host_elems(list(range(1, n + 1)))
        out = ctx.empty(C * n)
        for _ in range(3):
            lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(x), n, ctx.ptr(polys), C, d, ctx.ptr(out), ctx.stream())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 20; e0.record()
        for _ in range(K):
            lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(x), n, ctx.ptr(polys), C, d, ctx.ptr(out), ctx.stream())
        e1.record(); torch.cuda.synchronize()
        print(f"groups={G} n_out={n:3d}: {e0.elapsed_time(e1)/K*1e3:7.1f} us per launch", flush=True)
