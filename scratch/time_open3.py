"""The three launches of config 3's plan open (R1 encode, R1 decode + validate, R2 decode + validate) timed one by one with HIP events,
for A/B runs of library variants:   HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_<name>.so python scratch/time_open3.py [reps] [--nocheck]
Prints one line: the variant, the three averages (us, median of `reps` launches each, the launch alone between two events), the open
back to back, and whether the opened shares equal the secrets (variants that break the arithmetic on purpose say --nocheck)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
    check = "--nocheck" not in sys.argv
    n, t, B = 64, 21, 1 << 20
    d = t + 1
    C = (B + d - 1) // d
    ctx = Context.get(bench.BLS, 0)
    shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs(torch, ctx, n, t, B, False, seed=1000)
    order = np.random.Generator(np.random.PCG64(2024)).permutation(n).tolist()
    z, zc = order[:d], order[d:d + t]
    op = BatchOpen(bench.BLS, n, t, z=z, zc=zc, max_shares=B, device=0)
    if op.uses_fused_validate():
        op.set_fused_validate(True)
    r1_out, r2_msg, result = ctx.empty(n * C), ctx.empty(C), ctx.empty(B)
    legs = [lambda: op.r1_encode(shares0, out=r1_out), lambda: op.r1_decode(r1_cols, B, out=r2_msg), lambda: op.r2_decode(r2_cols, B, out=result)]

    def step():
        for leg in legs:
            leg()

    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        step()
    torch.cuda.synchronize()
    ok = op.ok() and bool(torch.equal(result, secrets))
    meds = []
    for leg in legs:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            step()          # the launch keeps its place in the open: same cache state as in the bench
            a.record()
            leg()
            b.record()
        torch.cuda.synchronize()
        meds.append(float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    name = os.path.basename(os.environ.get("HBMPC_HIP_LIB", "libhbmpc_hip.so"))
    print(f"{name:34s} encode {meds[0]:6.1f}  R1 {meds[1]:6.1f}  R2 {meds[2]:6.1f} us   open {dt * 1e6:6.1f} us = {B / dt / 1e9:5.2f} G shares/s   "
          f"{'bit-exact' if ok else ('(not checked)' if not check else 'MISMATCH')}", flush=True)
    if check and not ok:
        sys.exit(1)


main()
