"""Randomised differential test on the GPU: the int8 matrix-core kernels against the integer-VALU kernels
(and against exact Python integers for the encode) over random shapes, arrival orders and edge-heavy
inputs.  usage: python scratch/stress_open_paths.py [seconds] [seed]"""
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
EDGE = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, int("80" * 32, 16) % P, int("7f" * 32, 16) % P, int("ff00" * 16, 16) % P, 1 << 254, (1 << 254) - 1]


def rand_elems(rnd, count, edge_frac):
    out = []
    for _ in range(count):
        out.append(rnd.choice(EDGE) if rnd.random() < edge_frac else rnd.randrange(P))
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    ctx = Context.get(P)
    t_end = time.time() + budget
    trials = mm = 0
    as_np = lambda tns: tns.cpu().numpy().view(np.uint64)  # noqa: E731
    while time.time() < t_end:
        n = rnd.choice([1, 2, 3, 4, 5, 7, 8, 13, 16, 17, 22, 31, 32, 33, 40, 47, 48, 49, 63, 64])
        t = rnd.randrange(0, min(n, 32))
        if n ** t >= 127 * 256 ** 15:
            continue
        d = t + 1
        b = rnd.choice([1, d, d + 1, 16 * d, 16 * d + 1, 33 * d - 1, rnd.randrange(1, 4000)])
        c = (b + d - 1) // d
        order = list(range(n))
        rnd.shuffle(order)
        z, zc = order[:d], order[d : d + min(t, n - d)]
        op = BatchOpen(P, n, t, z=z, zc=zc, max_shares=b)
        if not op.uses_matrix_cores():
            continue
        shares = rand_elems(rnd, b, rnd.choice([0.0, 0.1, 0.9]))
        sh = ctx.upload_ints(shares)
        enc_m = op.r1_encode(sh)
        op.set_matrix_cores(False)
        enc_v = op.r1_encode(sh)
        assert np.array_equal(as_np(enc_m), as_np(enc_v)), ("encode", n, t, b)
        # exact check of a few outputs
        em = ctx.download_ints(enc_m)
        pad = shares + [0] * (c * d - b)
        for _ in range(5):
            i, k = rnd.randrange(n), rnd.randrange(c)
            want = sum(pow(i + 1, l, P) * pad[k * d + l] for l in range(d)) % P
            assert em[i * c + k] == want, ("encode exact", n, t, b, i, k)
        # the encode of `shares` is a consistent set of received columns
        bad = None
        if zc and rnd.random() < 0.5:
            bad = enc_m.clone()
            col, row = rnd.choice(zc), rnd.randrange(c)
            bad[col * c + row, rnd.randrange(4)] ^= 1 << rnd.randrange(60)
        outs = []
        for on in (True, False):
            op.set_matrix_cores(on)
            msg = op.r1_decode(enc_m, b)
            res = op.r2_decode(enc_m, b)
            assert op.ok(), ("validate", n, t, b, on)
            assert ctx.download_ints(res) == shares, ("decode", n, t, b, on)
            outs.append((as_np(msg).copy(), as_np(res).copy()))
            if bad is not None:
                op.r2_decode(bad, b)
                assert not op.ok(), ("corruption missed", n, t, b, on)
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), ("paths differ", n, t, b)
        trials += 1
        mm += 1
        del op
    torch.cuda.synchronize()
    print(f"stress: {trials} random opens agreed on both kernel families (seed {seed}, {budget:.0f} s)")


main()
