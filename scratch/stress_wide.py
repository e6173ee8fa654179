"""Randomised differential run focused on the shapes the full-size matrix-core kernel serves (omega points, large powers)."""
import random, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, int("80" * 32, 16) % P, int("7f" * 32, 16) % P, int("ff00" * 16, 16) % P, 1 << 254, (1 << 254) - 1]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
ctx = Context.get(P)
as_np = lambda tns: tns.cpu().numpy().view(np.uint64)
t_end = time.time() + budget
trials = fails = 0
while time.time() < t_end:
    n = rnd.choice([8, 16, 33, 64, 100, 128, 128, 100])
    t = rnd.randrange(3, min(n, 48))
    use_omega = rnd.random() < 0.4
    if not use_omega and n ** t < 127 * 256 ** 15:
        continue
    d = t + 1
    b = rnd.choice([16 * d + 1, 33 * d - 1, 64 * d, rnd.randrange(1, 6000)])
    c = (b + d - 1) // d
    order = list(range(n)); rnd.shuffle(order)
    z, zc = order[:d], order[d:d + min(t, n - d)]
    op = BatchOpen(P, n, t, z=z, zc=zc, use_omega_powers=use_omega, max_shares=b)
    if not op.uses_matrix_cores():
        continue
    shares = [rnd.choice(edge) if rnd.random() < rnd.choice([0.0, 0.1, 0.9]) else rnd.randrange(P) for _ in range(b)]
    sh = ctx.upload_ints(shares)
    enc_m = op.r1_encode(sh)
    op.set_matrix_cores(False)
    enc_v = op.r1_encode(sh)
    if not np.array_equal(as_np(enc_m), as_np(enc_v)):
        diff = (enc_m != enc_v).any(dim=1)
        idx = torch.nonzero(diff).flatten().tolist()
        rows = sorted(set(i // c for i in idx)); chunks = sorted(set(i % c for i in idx))
        op.set_matrix_cores(True)
        again = op.r1_encode(sh)
        print("ENCODE MISMATCH", n, t, b, use_omega, "C", c, "trial", trials, "cache", ctx.cache_entries(), len(idx), "wrong; rows", rows[:12], len(rows), "chunks", chunks[:12], len(chunks),
              "again equal:", bool(torch.equal(again, enc_v)), flush=True)
        op.set_matrix_cores(False)
        fails += 1
    for on in (True, False):
        op.set_matrix_cores(on)
        for rep in range(2):
            msg = op.r1_decode(enc_v, b)
            res = op.r2_decode(enc_v, b)
            ok = op.ok()
            good = ctx.download_ints(res) == shares
            if not (ok and good):
                print("FAIL", n, t, b, use_omega, "mc", on, "rep", rep, "ok", ok, "exact", good, flush=True); fails += 1
    trials += 1
    del op
print(f"stress_wide: {trials} opens, {fails} failures (seed {seed}, {budget:.0f} s)")
