"""GPU timeline of first-sight opens (config 3's shape): run under rocprofv3 --kernel-trace, then
    python scratch/first_sight_timeline.py --analyse <results.db>
prints, per open, the kernels in order with the idle gaps between them."""
import sys
if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # the last 40 opens: an open = k_mm8<...> (encode) then two k_mm8f
    seq = [(n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:28], s, e) for n, s, e in rows]
    first = sys.argv[3] if len(sys.argv) > 3 else "hb::k_mm8<"
    idx = [i for i, r in enumerate(seq) if r[0].startswith(first)]
    idx = idx[-41:]
    import collections
    if len(idx) < 3:
        print("no kernel named", first, "-- kernels seen:", collections.Counter(r[0] for r in seq[-400:]).most_common(12))
        sys.exit(1)
    npart = collections.Counter(b - a for a, b in zip(idx[:-1], idx[1:])).most_common(1)[0][0]
    tot = {}
    cnt = 0
    for a, b in zip(idx[:-1], idx[1:]):
        part = seq[a:b]
        if len(part) != npart:
            continue
        cnt += 1
        prev_end = None
        for j, (nm, s, e) in enumerate(part):
            key = f"{j}:{nm}"
            tot.setdefault(key, [0.0, 0.0])
            tot[key][0] += (e - s) / 1e3
            if prev_end is not None:
                tot[key][1] += (s - prev_end) / 1e3
            prev_end = e
        tot.setdefault("open", [0.0, 0.0])
        tot["open"][0] += (seq[b][1] - part[0][1]) / 1e3
    print(f"{cnt} opens; per open: kernel us (idle gap before it us)")
    for k, (d, g) in tot.items():
        print(f"  {k:<40} {d / cnt:8.1f}  ({g / cnt:6.1f})")
    sys.exit(0)
import gc
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder
P = bench.BLS
n, t, B = 64, 21, 1 << 20
OMEGA = len(sys.argv) > 1 and sys.argv[1] == "omega"
if len(sys.argv) > 1 and sys.argv[1] == "cfg5":
    n, t, B, OMEGA = 256, 85, (1 << 22) // 8, True
d = t + 1
C = (B + d - 1) // d
ctx = Context.get(P, 0)
shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs_light(torch, ctx, n, t, B, OMEGA, seed=1000)
op = BatchOpen(P, n, t, max_shares=B, device=0, use_omega_powers=OMEGA)
r1_out = ctx.empty(n * C)
r1v, r2v = r1_cols.view(n, C, 4), r2_cols.view(n, C, 4)
rng = np.random.Generator(np.random.PCG64(77))
MODE = next((a for a in sys.argv[1:] if a in ("wait", "defer", "early")), "defer")
def first_sight(o1, o2):
    op.r1_encode(shares0, out=r1_out)
    if MODE == "wait":             # rounds 4-5: every quorum's add() waits for its verdict
        outs = []
        for order_, cols_, want_ in ((o1, r1v, "constant"), (o2, r2v, "all")):
            dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, device=0, columns=cols_, want=want_, use_omega_powers=OMEGA)
            for idx in order_:
                dec.add(idx)
                if dec.done():
                    break
            outs.append(dec.get_results()[0])
        return outs
    mk = lambda cols_, want_, busy_: DeviceIncrementalDecoder(P, n, t, batch_size=C, device=0, columns=cols_, want=want_, use_omega_powers=OMEGA, defer_verdict=True, stream_busy=busy_)
    dec1, dec2 = mk(r1v, "constant", "--r1-in-order" not in sys.argv), None
    for idx in o1:
        dec1.add(idx)
        if dec1.pending():
            dec2 = mk(r2v, "all", MODE == "early")
            if MODE == "early":
                for j in o2:
                    dec2.add(j)
                    if dec2.pending():
                        break
        if dec1.done():
            break
    m1 = dec1.get_results()[0]
    if dec2 is None:
        dec2 = mk(r2v, "all", False)
    if not dec2.pending():
        for idx in o2:
            dec2.add(idx)
            if dec2.done():
                break
    return [m1, dec2.get_results()[0]]
orders = [(rng.permutation(n).tolist(), rng.permutation(n).tolist()) for _ in range(120)]
for o in orders[:10]:
    first_sight(*o)
gc.collect(); gc.freeze()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for o in orders[10:]:
    res = first_sight(*o)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 110
print(f"first-sight open [{MODE}]: {dt*1e6:.1f} us = {B/dt/1e9:.2f} G shares/s")
assert torch.equal(res[1].reshape(-1, 4)[:B], secrets)
if "--timing" in sys.argv:          # the timing build (scratch/build_variant2.sh timing ... -DHB_MM8_TIMING): phases of the LAST k_mm8f launch (R2)
    import ctypes
    fs = ctx.lib.hb_debug_fs_timing
    fs.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert fs(buf.ctypes.data, buf.size) == 0
    tt = buf.reshape(256 * 8, 8).astype(np.float64)
    names = ["prologue+unit 0 scaled", "barrier wait", "scale next unit", "MFMA half 0", "park half 0", "MFMA half 1", "words+reduce+store/compare", "lgkm wait before barrier"]
    tot = tt.sum(axis=1)
    print(f"R2 launch of the last open: per-wave total ticks mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
    for k, nm in enumerate(names):
        print(f"  {nm:28s} {tt[:, k].mean():9.0f}  (min {tt[:, k].min():.0f}, max {tt[:, k].max():.0f})")
