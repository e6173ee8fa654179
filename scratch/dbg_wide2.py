import random, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P)
rnd = random.Random(3)
found = 0
t0 = time.time()
it = 0
while time.time() - t0 < 100 and found < 12:
    it += 1
    n, t, b, om = rnd.choice([(100, 21, 353, False), (64, 28, 465, False), (33, 12, 832, True), (128, 3, 256, True), (128, 41, 2688, False)])
    d = t + 1; c = (b + d - 1) // d
    order = list(range(n)); rnd.shuffle(order)
    z, zc = order[:d], order[d:d + min(t, n - d)]
    op = BatchOpen(P, n, t, z=z, zc=zc, use_omega_powers=om, max_shares=b)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, int("80" * 32, 16) % P, int("7f" * 32, 16) % P, int("ff00" * 16, 16) % P, 1 << 254, (1 << 254) - 1]
    frac = rnd.choice([0.1, 0.9])
    vals = [rnd.choice(edge) if rnd.random() < frac else rnd.randrange(P) for _ in range(b)]
    sh = ctx.upload_ints(vals)
    enc_m = op.r1_encode(sh).clone()
    enc_m2 = op.r1_encode(sh).clone()
    op.set_matrix_cores(False)
    enc_v = op.r1_encode(sh)
    for name, e in (("first", enc_m), ("second", enc_m2)):
        diff = (e != enc_v).any(dim=1)
        if bool(diff.any().item()):
            idx = torch.nonzero(diff).flatten().tolist()
            rows = sorted(set(i // c for i in idx)); chunks = sorted(set(i % c for i in idx))
            print(f"it {it} {name} launch: n={n} t={t} b={b} om={om} C={c}: {len(idx)} wrong elements; rows {rows[:20]} (count {len(rows)}); chunks {chunks[:20]} (count {len(chunks)})", flush=True)
            # how wrong: print one
            i0 = idx[0]
            ci, ri = i0 % c, i0 // c
            print("   chunk", ci, "row", ri, "inputs", [hex(v)[:20] for v in (vals + [0] * d)[ci * d:(ci + 1) * d]][:8], flush=True)
            print("   got", [hex(v & 0xFFFFFFFFFFFFFFFF) for v in e[i0].tolist()], "want", [hex(v & 0xFFFFFFFFFFFFFFFF) for v in enc_v[i0].tolist()], flush=True)
            found += 1
    del op
print("iterations", it, "found", found)
