"""Randomised differential run of hb_quick_interp_check (matrix built on the device) against the plan-based open (BatchOpen.r2_decode: tables built
on the host): random n / t / arrival sets / compared sets / point policies / batch sizes / chunk ranges; coefficients bit for bit, the disagreement
flag and the FIRST disagreeing chunk after corrupting random compared symbols.   usage: python scratch/stress_quick.py [seconds] [seed]"""
import random, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.device import BatchOpen
from honeybadgermpc_amd.field import GF
from honeybadgermpc_amd.polynomial import EvalPoint
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
ctx = Context.get(P)
gen = torch.Generator(device='cuda'); gen.manual_seed(seed)
INT_MAX = (1 << 31) - 1
t_end = time.time() + budget
runs = fails = 0
while time.time() < t_end:
    n = rnd.choice([7, 8, 16, 22, 31, 48, 64, 64, 100, 128, 200, 256])
    t = rnd.randrange(3, min((n - 1) // 3, 85) + 1) if (n - 1) // 3 >= 3 else (n - 1) // 2
    d = t + 1
    if d < 4 or d > 128:
        continue
    use_omega = rnd.random() < 0.4
    c = rnd.choice([1, 15, 16, 17, 100, 777, rnd.randrange(1, 3000)])
    order = list(range(n)); rnd.shuffle(order)
    z = order[:d]
    nc = rnd.choice([0, 1, t, min(n - d, t + 5), min(n - d, 256)])
    zc = order[d:d + nc]
    point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
    x = [point(i).value for i in range(n)]
    xh = ctx.host_elems(x)
    coef = torch.randint(-(1 << 63), (1 << 63) - 1, (c * d, 4), dtype=torch.int64, device='cuda', generator=gen); coef[:, 3] &= (1 << 61) - 1
    plan = BatchOpen(P, n, t, z=z, zc=zc[:t], use_omega_powers=use_omega, max_shares=c * d)
    cols = plan.r1_encode(coef).clone()                       # [n][c] consistent columns
    want = plan.r2_decode(cols, c * d)                        # plan-based decode of the same columns = coef
    assert plan.ok() and torch.equal(want, coef)
    lo = rnd.choice([0, 0, rnd.randrange(0, c)])
    hi = rnd.choice([c, c, rnd.randrange(lo, c + 1)])
    bad_chunks = []
    data = cols
    if zc and rnd.random() < 0.6:
        data = cols.clone().view(n, c, 4)
        for _ in range(rnd.randrange(1, 4)):
            m = rnd.randrange(c)
            data[rnd.choice(zc), m, rnd.randrange(4)] ^= 1 << rnd.randrange(60)
            bad_chunks.append(m)
        # what actually differs (two flips of the same bit cancel)
        differs = (data != cols.view(n, c, 4)).any(dim=2).any(dim=0)
        bad_chunks = torch.nonzero(differs).flatten().tolist()
        data = data.view(n * c, 4)
    out = torch.zeros((c * d, 4), dtype=torch.int64, device='cuda')
    status = torch.tensor([0, INT_MAX], dtype=torch.int32, device='cuda')
    za, zca = np.array(z, dtype=np.int32), np.array(zc if zc else [0], dtype=np.int32)
    rc = ctx.lib.hb_quick_interp_check(ctx.h, np_ptr(xh), n, np_ptr(za), d, np_ptr(zca), len(zc), ctx.ptr(data), c, lo, hi, ctx.ptr(out), ctx.ptr(status), ctx.stream())
    if rc == 3:      # UNSUPPORTED shape: fine, callers fall back
        continue
    ctx.check(rc, "quick")
    flag, first = status.tolist()
    inside = sorted(m for m in bad_chunks if lo <= m < hi)
    exp_flag, exp_first = (1, inside[0] - lo) if inside else (0, INT_MAX)
    good = torch.equal(out[lo * d:hi * d], coef[lo * d:hi * d]) and not bool(out[:lo * d].any()) and not bool(out[hi * d:].any())
    if not good or (flag, first) != (exp_flag, exp_first):
        fails += 1
        print("FAIL", n, t, c, use_omega, "nc", len(zc), "range", lo, hi, "coeffs ok", good, "status", (flag, first), "expected", (exp_flag, exp_first), flush=True)
    runs += 1
    del plan
print(f"stress_quick: {runs} launches, {fails} failures (seed {seed}, {budget:.0f} s)")
