import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen

p = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
ctx = Context.get(p, 0)
n, t = 24, 5
d = t + 1
x = list(range(1, n + 1))
for c in (300, 700):
  for src in ("randint", "python"):
    rnd = random.Random(3)
    if src == "python":
        coefs = [rnd.randrange(p) for _ in range(c * d)]
        coef = ctx.upload_ints(coefs)
    else:
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        coef = torch.randint(-(1 << 63), (1 << 63) - 1, (c * d, 4), dtype=torch.int64, device="cuda", generator=gen)
        coef[:, 3] &= (1 << 61) - 1
        coefs = ctx.download_ints(coef)
    want = oracle.vandermonde_batch_evaluate(x, [coefs[i * d:(i + 1) * d] for i in range(c)], p)
    want_pm = [want[k][j] for j in range(n) for k in range(c)]
    V = BatchOpen(p, n, t, max_shares=c * d)
    got = ctx.download_ints(V.r1_encode(coef))
    bad = [i for i in range(len(got)) if got[i] != want_pm[i]]
    print(c, src, "encode", got == want_pm, len(bad), bad[:5])
    cols = ctx.upload_ints(want_pm)
    r = random.Random(1)
    for it in range(3):
        z = r.sample(range(n), d)
        zc = [j for j in r.sample(range(n), n) if j not in z][:t]
        for mc, fused in ((True, True), (True, False), (False, False)):
            op = BatchOpen(p, n, t, z=z, zc=zc, max_shares=c * d)
            op.set_matrix_cores(mc)
            op.set_fused_validate(fused)
            res = op.r2_decode(cols, c * d)
            fine = op.ok()
            g = ctx.download_ints(res)
            bad = [i for i in range(len(g)) if g[i] != coefs[i]]
            print(c, src, "decode mc", mc, "fused", fused, fine, g == coefs, len(bad), bad[:4], z)
