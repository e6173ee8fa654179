"""Host-side timeline of a first-sight open (config 3's shape, deferred verdicts): when, after the open's start, each step of the host loop RETURNS.
Averages over 200 opens; us."""
import gc, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder
P = bench.BLS
n, t, B = 64, 21, 1 << 20
d = t + 1
C = (B + d - 1) // d
ctx = Context.get(P, 0)
shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs_light(torch, ctx, n, t, B, False, seed=1000)
op = BatchOpen(P, n, t, max_shares=B, device=0)
r1_out = ctx.empty(n * C)
r1v, r2v = r1_cols.view(n, C, 4), r2_cols.view(n, C, 4)
rng = np.random.Generator(np.random.PCG64(77))
mk = lambda cols_, want_, busy_=False: DeviceIncrementalDecoder(P, n, t, batch_size=C, device=0, columns=cols_, want=want_, defer_verdict=True, stream_busy=busy_)
acc = {}
def open_once(o1, o2, rec):
    t0 = time.perf_counter()
    def T(name):
        if rec:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    op.r1_encode(shares0, out=r1_out); T("01 encode enqueued")
    dec1 = mk(r1v, "constant", True); T("02 R1 decoder made")
    for k, idx in enumerate(o1):
        dec1.add(idx)
        if k == d - 2: T("03 R1 add #21")
        if k == d - 1: T("04 R1 add #22 (first half enqueued)")
        if dec1.pending():
            break
    T("05 R1 add #43 (decode + validate enqueued)")
    dec2 = mk(r2v, "all"); T("06 R2 decoder made")
    dec1.done(); T("07 R1 verdict in")
    m1 = dec1.get_results()[0]; T("08 R1 results")
    for k, idx in enumerate(o2):
        dec2.add(idx)
        if k == d - 1: T("09 R2 add #22 (first half enqueued)")
        if dec2.pending():
            break
    T("10 R2 add #43 (decode + validate enqueued)")
    dec2.done(); T("11 R2 verdict in")
    r = dec2.get_results()[0]; T("12 R2 results")
    return r
orders = [(rng.permutation(n).tolist(), rng.permutation(n).tolist()) for _ in range(210)]
for o in orders[:10]:
    open_once(*o, False)
gc.collect(); gc.freeze(); torch.cuda.synchronize()
for o in orders[10:]:
    res = open_once(*o, True)
torch.cuda.synchronize()
prev = 0.0
for k in sorted(acc):
    v = acc[k] / 200 * 1e6
    print(f"  {k:<46} {v:7.1f}  (+{v - prev:5.1f})")
    prev = v
assert torch.equal(res.reshape(-1, 4)[:B], secrets)
