"""Device and host memory over many opens and decoders (plan API with a fresh arrival set each time through the plan cache, first-sight decoders fault-free and
under attack, robust batches): free HBM and the process's RSS at intervals -- a leak shows as a slope.  python scratch/leak_check.py [seconds]"""
import os, random, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder, cached_batch_open
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(5)
ctx = Context.get(P)
gen = torch.Generator(device='cuda'); gen.manual_seed(3)
def rand(count):
    v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device='cuda', generator=gen); v[:, 3] &= (1 << 61) - 1; return v
def rss():
    with open('/proc/self/statm') as f:
        return int(f.read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2**20
def free_hbm():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20
shapes = [(64, 21, 1 << 14, False), (16, 5, 1 << 12, True), (100, 33, 1 << 12, False), (256, 85, 1 << 13, True)]
data = {}
for n, t, B, om in shapes:
    d = t + 1; C = (B + d - 1) // d
    enc = BatchOpen(P, n, t, use_omega_powers=om, max_shares=C * d)
    cols = enc.r1_encode(rand(C * d)).view(n, C, 4).clone()
    bad = cols.clone()
    for i in range(t):
        bad[i] = rand(C)
    data[(n, t)] = (C, om, cols, bad)
    del enc
t_end = time.time() + budget
it = 0
marks = []
while time.time() < t_end:
    n, t, B, om = shapes[it % len(shapes)]
    C, om, cols, bad = data[(n, t)]
    order = list(range(n)); rnd.shuffle(order)
    kind = it % 3
    src = cols if kind == 0 else bad
    if kind == 2:
        order = list(range(t)) + [i for i in order if i >= t]          # liars first
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=C, columns=src, use_omega_powers=om)
    for idx in order:
        dec.add(idx)
        if dec.done():
            break
    res, errs = dec.get_results()
    assert res is not None and errs <= set(range(t)) and (kind != 0 or not errs), (n, t, kind, errs)   # (a decode can finish before the last liar has arrived)
    z = order[: t + 1]; zc = order[t + 1: 2 * t + 1]
    op = cached_batch_open(P, n, t, tuple(z), tuple(zc), use_omega_powers=om, max_shares=C * (t + 1))
    op.r2_decode(cols.view(-1, 4), C * (t + 1))
    assert op.ok()
    del dec, op, res
    it += 1
    if it % 200 == 0:
        marks.append((it, round(free_hbm()), round(rss())))
        print(f"{it} decoders + opens: free HBM {marks[-1][1]} MiB, RSS {marks[-1][2]} MiB", flush=True)
if len(marks) >= 3:
    print(f"leak_check: {it} iterations; free HBM {marks[0][1]} -> {marks[-1][1]} MiB, RSS {marks[0][2]} -> {marks[-1][2]} MiB between iteration {marks[0][0]} and {marks[-1][0]}")
