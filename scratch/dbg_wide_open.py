import random, sys
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P)
rnd = random.Random(1)
for (n, t, b, om) in [(128, 26, 890, False), (128, 26, 27 * 16, False), (128, 26, 27 * 17, False), (128, 26, 27 * 5, False), (100, 33, 34 * 40, False), (64, 21, 22 * 50, True), (128, 42, 43 * 33, False), (128, 26, 890, False)]:
    d = t + 1; c = (b + d - 1) // d
    order = list(range(n)); rnd.shuffle(order)
    z, zc = order[:d], order[d:d + min(t, n - d)]
    op = BatchOpen(P, n, t, z=z, zc=zc, use_omega_powers=om, max_shares=b)
    shares = [rnd.randrange(P) for _ in range(b)]
    sh = ctx.upload_ints(shares)
    enc = op.r1_encode(sh)
    for on in (True, False):
        op.set_matrix_cores(on)
        res = op.r2_decode(enc, b)
        ok = op.ok()
        good = ctx.download_ints(res) == shares
        print(n, t, b, om, "C", c, "mc", on, "uses", op.uses_matrix_cores(), "ok", ok, "decode exact", good, flush=True)
