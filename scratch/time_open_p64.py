"""The three launches of config 3's open over the 64-bit prime timed one by one (HIP events), for A/B runs of library variants:
   HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_<name>.so python scratch/time_open_p64.py [reps] [--nocheck]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from honeybadgermpc_amd._capi import Context, HbView, np_ptr  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
    check = "--nocheck" not in sys.argv
    n, t, B = 64, 21, 1 << 20
    d = t + 1
    C = (B + d - 1) // d
    P = bench.P64
    ctx = Context.get(P, 0, 1)
    lib = ctx.lib
    xh = ctx.host_elems(list(range(1, n + 1)))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(6400)

    def rnd(count):
        return ctx.reduce_(torch.randint(-(1 << 63), (1 << 63) - 1, (count, 1), dtype=torch.int64, device="cuda", generator=gen))

    secrets, shares0 = rnd(B), rnd(B)
    pad = C * d - B
    sec_pad = secrets if not pad else torch.cat([secrets, torch.zeros((pad, 1), dtype=torch.int64, device="cuda")])
    os.environ_backup = None
    V = ctypes.c_void_p()
    ctx.check(lib.hb_vand_matrix_create(ctx.h, np_ptr(xh), n, d, ctypes.byref(V), ctx.stream()), "V")
    r2_cols = ctx.empty(n * C)
    ctx.check(lib.hb_matvec(ctx.h, V, ctx.ptr(sec_pad), HbView(d, 1), None, ctx.ptr(r2_cols), HbView(1, C), C, ctx.stream()), "r2cols")
    g = rnd(d * C)
    g[:C] = r2_cols[:C]
    r1_cols = ctx.empty(n * C)
    ctx.check(lib.hb_matvec(ctx.h, V, ctx.ptr(g), HbView(1, C), None, ctx.ptr(r1_cols), HbView(1, C), C, ctx.stream()), "r1cols")
    torch.cuda.synchronize()
    order = np.random.Generator(np.random.PCG64(2024)).permutation(n).tolist()
    z, zc = order[:d], order[d:d + t]
    op = BatchOpen(P, n, t, z=z, zc=zc, max_shares=B, device=0)
    r1_out, r2_msg, result = ctx.empty(n * C), ctx.empty(C), ctx.empty(B)
    legs = [lambda: op.r1_encode(shares0, out=r1_out), lambda: op.r1_decode(r1_cols, B, out=r2_msg), lambda: op.r2_decode(r2_cols, B, out=result)]

    def step():
        for leg in legs:
            leg()

    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        step()
    torch.cuda.synchronize()
    ok = bool(torch.equal(result, secrets)) and (not check or op.ok())
    meds = []
    for leg in legs:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); b.record()
        torch.cuda.synchronize()
        for a, b in evs:
            step()
            a.record()
            leg()
            b.record()
        torch.cuda.synchronize()
        meds.append(float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3)
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    name = os.path.basename(os.environ.get("HBMPC_HIP_LIB", "libhbmpc_hip.so")) + (" NO_MFMA" if os.environ.get("HB_NO_MFMA") else "")
    print(f"{name:34s} encode {meds[0]:6.1f}  R1 {meds[1]:6.1f}  R2 {meds[2]:6.1f} us   open {dt * 1e6:6.1f} us = {B / dt / 1e9:5.2f} G shares/s   "
          f"{'bit-exact' if ok else ('(not checked)' if not check else 'MISMATCH')}", flush=True)


main()
