"""Where a wave of k_mm8w spends its time: needs the -DHB_MM8_TIMING build
(scratch/build_variant.sh timing -DHB_MM8_TIMING; HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_timing.so)."""
import ctypes, random, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context, HbView, np_ptr
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
NAMES = ["prologue", "asm pass", "DMA issue", "bookkeeping", "vmcnt wait", "barrier", "reload", "-"]
ctx = Context.get(P); lib = ctx.lib
rnd = random.Random(3)
g = torch.Generator(device='cuda'); g.manual_seed(1)
for nn, dd, CC in [(64, 22, 47663), (22, 22, 47663), (86, 86, 6097)]:
    M = [[rnd.randrange(P) for _ in range(dd)] for _ in range(nn)]
    h = ctypes.c_void_p()
    ctx.check(lib.hb_matrix_from_host(ctx.h, np_ptr(ctx.host_elems([v for r in M for v in r])), nn, dd, ctypes.byref(h), ctx.stream()), "from_host")
    x = torch.randint(-(1 << 63), (1 << 63) - 1, (CC * dd, 4), dtype=torch.int64, device='cuda', generator=g); x[:, 3] &= (1 << 61) - 1
    o = ctx.empty(CC * nn)
    def run():
        ctx.check(lib.hb_matvec(ctx.h, h, ctx.ptr(x), HbView(1, CC), None, ctx.ptr(o), HbView(1, CC), CC, ctx.stream()), "mv")
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    fn = lib.hb_debug_mm8w_timing; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert fn(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(1024, 8).astype(np.float64)
    tot = t.sum(axis=1)
    print(f"{nn}x{dd} C={CC}: kernel {dt * 1e6:.1f} us; per-wave ticks mean {tot.mean():.0f} max {tot.max():.0f} -> {tot.mean() / (dt * 1e6):.1f} ticks/us")
    for k, nm in enumerate(NAMES[:7]):
        col = t[:, k]
        print(f"  {nm:14s} {col.mean():9.0f} ticks/wave {100 * col.mean() / tot.mean():5.1f} %  = {col.mean() / 100:8.1f} us  (min {col.min():.0f} max {col.max():.0f})")

# the fused decode + validate launches of an open (config 3): R1 = 22 rows, R2 = 43 rows, CHECK kernel
from honeybadgermpc_amd.device import BatchOpen
import random as _r
n_, t_ = 64, 21
d_ = t_ + 1
B_ = 1 << 20
order = list(range(n_)); _r.Random(7).shuffle(order)
op = BatchOpen(P, n_, t_, z=order[:d_], zc=order[d_:d_ + t_], max_shares=B_)
sh = torch.randint(0, 1 << 62, (B_, 4), dtype=torch.int64, device='cuda', generator=g)
cols = op.r1_encode(sh)
for name, run_ in (("R1 fused 22x22", lambda: op.r1_decode(cols, B_)), ("R2 fused 43x22", lambda: op.r2_decode(cols, B_))):
    for _ in range(3): run_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run_()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    assert op.ok()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(1024, 8).astype(np.float64); t[:, 7] = 0
    t = t[t.sum(axis=1) > 0]
    tot = t.sum(axis=1)
    print(f"{name}: kernel {dt * 1e6:.1f} us; per-wave ticks mean {tot.mean():.0f} max {tot.max():.0f}")
    for k, nm in enumerate(NAMES[:7]):
        col = t[:, k]
        print(f"  {nm:14s} {col.mean():9.0f} ticks/wave {100 * col.mean() / tot.mean():5.1f} %  (min {col.min():.0f} max {col.max():.0f})")
