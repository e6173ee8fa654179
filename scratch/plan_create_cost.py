"""BatchOpen plan creation time, with and without the fused decode + validate matrices (HB_NO_FUSED_VALIDATE=1)."""
import os, sys, time, random
import torch
sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P)
rnd = random.Random(1)
for n, t, omega in [(64, 21, False), (64, 21, True), (256, 85, True), (16, 5, True)]:
    d = t + 1
    times = []
    for rep in range(6):
        order = list(range(n)); rnd.shuffle(order)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        op = BatchOpen(P, n, t, z=order[:d], zc=order[d:d + t], use_omega_powers=omega, max_shares=1 << 16)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        del op
    print(f"n={n} t={t} omega={omega} fused={'off' if os.environ.get('HB_NO_FUSED_VALIDATE') else 'on'}: plan creation {min(times[1:]):.2f} ms (first {times[0]:.2f})")
