"""Where a wave of k_mm8 spends its cycles: needs the -DHB_MM8_TIMING build of the library
(scratch/build_variant.sh timing -DHB_MM8_TIMING; HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_timing.so)."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
NAMES = ["prologue", "barrier wait", "DMA issue", "MFMA half 0", "park half 0", "MFMA half 1", "vmcnt wait", "words+reduce+store"]


def main():
    n_out, d = 64, 22
    ctx = Context.get(P)
    lib = ctx.lib
    xh = ctx.host_elems(list(range(1, n_out + 1)))
    h = ctypes.c_void_p()
    assert lib.hb_debug_mm8_create(ctx.h, xh.ctypes.data, n_out, d, ctypes.byref(h)) == 0
    chunks = (1 << 20) // d + 1
    x = ctx.empty(chunks * d)
    x.random_(0, 1 << 62)
    out = ctx.empty(chunks * n_out)

    def run():
        rc = lib.hb_debug_mm8_apply(ctx.h, h, ctx.ptr(x), d, 1, chunks * d, ctx.ptr(out), 1, chunks, chunks * n_out, chunks, None, None)
        assert rc == 0

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    fn = lib.hb_debug_mm8_timing
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert fn(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(2048, 8).astype(np.float64)
    tot = t.sum(axis=1)
    print(f"kernel {dt * 1e6:.1f} us; per-wave total ticks: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}  -> {tot.mean() / (dt * 1e6):.1f} ticks/us")
    passes = 2979 / 512
    for k, nm in enumerate(NAMES):
        col = t[:, k]
        print(f"  {nm:22s} {col.mean():9.0f} ticks/wave  {100 * col.mean() / tot.mean():5.1f} %   per pass {col.mean() / passes:7.0f}   (min {col.min():.0f}, max {col.max():.0f})")


main()
