"""Randomised differential test on the GPU for opens over word-size primes: the matrix-core kernel k_mv64m (hb_narrow.hip) against the generic
one-limb kernels (set_matrix_cores(False)) and against exact Python integers, over random primes of 42 .. 64 bits, shapes, arrival orders,
batch sizes and edge-heavy inputs; a lie planted in a compared column must be refused by both.
usage: python scratch/stress_narrow.py [seconds] [seed]"""
import random
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen  # noqa: E402


def is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    primes = [(1 << 64) - 59, 0xFFFFFFFF00000001, (1 << 61) - 1, (1 << 41) + 27, (1 << 63) + 29]
    primes = [q for q in primes if is_prime(q)]
    while len(primes) < 12:
        q = rnd.getrandbits(rnd.randrange(42, 65)) | 1 | (1 << 41)
        if is_prime(q):
            primes.append(q)
    t_end = time.time() + budget
    trials = n_om = 0
    as_np = lambda tns: tns.cpu().numpy().view(np.uint64)  # noqa: E731
    while time.time() < t_end:
        p = rnd.choice(primes)
        ctx = Context.get(p)
        edge = [0, 1, p - 1, p - 2, (p - 1) // 2, 0x8080808080808080 % p, 0x7f7f7f7f7f7f7f7f % p, 0xff00ff00ff00ff00 % p, 0x80 % p, (1 << 63) % p]
        n = rnd.choice([2, 3, 4, 5, 7, 8, 13, 16, 17, 22, 31, 32, 33, 40, 47, 48, 49, 63, 64, 65, 80, 100])
        t = rnd.randrange(0, min(n, 24))
        d = t + 1
        b = rnd.choice([1, d, d + 1, 16 * d, 16 * d + 1, 33 * d - 1, rnd.randrange(1, 6000)])
        c = (b + d - 1) // d
        order = list(range(n))
        rnd.shuffle(order)
        z, zc = order[:d], order[d : d + min(t, n - d)]
        frac = rnd.choice([0.0, 0.1, 0.9])
        pick = lambda: rnd.choice(edge) if rnd.random() < frac else rnd.randrange(p)  # noqa: E731
        shares = [pick() for _ in range(b)]
        polys = [[pick() for _ in range(d)] for _ in range(c)]
        order2 = 2 * (n if n & (n - 1) == 0 else 2 ** n.bit_length())
        om = (p - 1) % order2 == 0 and rnd.random() < 0.5          # plans at omega powers where the prime has the roots
        x = BatchOpen(p, n, t, z=z, zc=zc, use_omega_powers=om, max_shares=b).x if om else list(range(1, n + 1))
        cols = [[sum(pow(x[j], l, p) * polys[k][l] for l in range(d)) % p for k in range(c)] for j in range(n)]
        flat = ctx.upload_ints([v for col in cols for v in col])
        sh = ctx.upload_ints(shares)
        want_res = [v for row in polys for v in row][:b]
        pad = shares + [0] * (c * d - b)
        outs = []
        for cores in (True, False):
            op = BatchOpen(p, n, t, z=z, zc=zc, use_omega_powers=om, max_shares=b)
            if om and op.x != x:
                break                 # (a prime whose seeded root is not primitive draws again unseeded, as the reference does: other points, skip)
            if not cores:
                op.set_matrix_cores(False)
            enc = ctx.download_ints(op.r1_encode(sh))
            for _ in range(6):
                i, k = rnd.randrange(n), rnd.randrange(c)
                assert enc[i * c + k] == sum(pow(x[i], l, p) * pad[k * d + l] for l in range(d)) % p, ("encode exact", p, n, t, b, i, k, cores)
            msg = ctx.download_ints(op.r1_decode(flat, b))
            assert msg == [polys[k][0] for k in range(c)], ("r1", p, n, t, b, cores)
            res = ctx.download_ints(op.r2_decode(flat, b))
            assert res == want_res, ("r2", p, n, t, b, cores)
            assert op.ok(), ("ok", p, n, t, b, cores)
            outs.append(enc)
            if zc:
                lied = [list(col) for col in cols]
                j, k = rnd.choice(zc), rnd.randrange(c)
                lied[j][k] = (lied[j][k] + 1 + rnd.randrange(p - 1)) % p
                op.r2_decode(ctx.upload_ints([v for col in lied for v in col]), b)
                assert not op.ok(), ("lie accepted", p, n, t, b, j, k, cores)
        if len(outs) == 2:
            assert outs[0] == outs[1], ("encode differs", p, n, t, b)
        trials += 1
        n_om += 1 if om and len(outs) == 2 else 0
    print(f"stress_narrow: {trials} random opens ({n_om} at omega powers) over {len(primes)} primes (42 .. 64 bits), k_mv64m == generic kernels == exact integers, every planted lie refused; 0 differences")


main()
