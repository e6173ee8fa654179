#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (runs only where /root/reference exists).

Randomised differential run of the oracle's Gao decoder (oracle/hbmpc_oracle.c: partial_gcd + gao_interpolate) against the same recurrence
executed with the REFERENCE's own Polynomial class (oracle/gen_golden.py: reference_polynomial_gao -- its interpolate, __mul__, __sub__,
__divmod__; polynomial.py:85-108, 202-234): coefficients, the un-normalised cofactor and the (None, None) decisions.  Small fields make
the degenerate Euclid steps common (remainders whose degree drops by more than one, vanishing remainders and cofactor coefficients), which
tests/golden/gao_cofactor.json (54 words over BLS12-381's r) can hardly contain.

    python oracle/diff_gao_vs_reference_polynomial.py [seconds] [seed]
"""
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

import oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
FIELDS = [gg.BLS, 13, 53, 257, 65537, (1 << 61) - 1]
gaos = {p: gg.reference_polynomial_gao(p) for p in FIELDS}
t_end = time.time() + budget
words = fails = refused = 0
while time.time() < t_end:
    p = rnd.choice(FIELDS)
    n = rnd.randrange(2, min(p - 1, 24) + 1)
    k = rnd.randrange(1, n + 1)
    xs = rnd.sample(range(1, min(p, 6 * n)), n) if rnd.random() < 0.4 else list(range(1, n + 1))
    kind = rnd.random()
    if kind < 0.15:
        msg = [0] * k
    elif kind < 0.4:
        keep = rnd.randrange(k + 1)
        msg = [rnd.randrange(p) for _ in range(keep)] + [0] * (k - keep)
    else:
        msg = [rnd.randrange(p) for _ in range(k)]
    ys = [sum(c * pow(x, e, p) for e, c in enumerate(msg)) % p for x in xs]
    n_er = rnd.randrange(0, max(1, n - k)) if rnd.random() < 0.3 else 0
    for pos in rnd.sample(range(n), n_er):
        ys[pos] = None
    live = [i for i in range(n) if ys[i] is not None]
    radius = max(0, (len(live) - k) // 2)
    ne = min(len(live), rnd.choice([0, radius, rnd.randrange(radius + 1), radius + 1, radius + 2, rnd.randrange(len(live) + 1)]))
    for pos in rnd.sample(live, ne):
        ys[pos] = (ys[pos] + rnd.randrange(1, p)) % p
    if len(live) < 1:
        continue
    want = gaos[p](xs, ys, k)
    got = oracle.gao_interpolate(xs, ys, k, p)
    got = (got[0], got[1]) if got[0] is not None else (None, None)
    if (want[0] is None) != (got[0] is None) or (want[0] is not None and (list(want[0]) != list(got[0]) or list(want[1]) != list(got[1]))):
        fails += 1
        print("FAIL", p, n, k, "x", xs, "y", ys, "reference-polynomial", want, "oracle", got, flush=True)
    words += 1
    refused += want[0] is None
print(f"diff_gao_vs_reference_polynomial: {words} words ({refused} refused by both), {fails} differences (seed {seed}, {budget:.0f} s)")
