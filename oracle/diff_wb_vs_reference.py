#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (runs only where /root/reference exists).

Randomised differential run of the oracle's Welch-Berlekamp decoder (oracle/hbmpc_oracle.c, wb_decode) against the REFERENCE's own
make_wb_encoder_decoder(...).decode (reed_solomon_wb.py:47-153, pure Python): coefficients (trailing zeros stripped), or the text of the
exception it raises ("No solution", "found no divisors!", the bare assertion on 2t + 1 + c <= n).  Random and structured messages (zero,
constant, leading zeros), erasures, error counts up to and beyond the radius, small fields and BLS12-381's r.

    python oracle/diff_wb_vs_reference.py [seconds] [seed]
"""
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

import oracle  # noqa: E402
import logging  # noqa: E402

logging.disable(logging.CRITICAL)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
FIELDS = [gg.BLS, 53, 257, 65537]
t_end = time.time() + budget
words = fails = refused = 0
codecs = {}
while time.time() < t_end:
    p = rnd.choice(FIELDS)
    n = rnd.randrange(3, 17)
    k = rnd.randrange(1, (n - 1) // 2 + 2)
    t = k - 1
    if 2 * t + 1 > n:
        continue
    key = (p, n, k)
    if key not in codecs:
        codecs[key] = gg.make_wb_encoder_decoder(n, k, p)
    enc, dec, _ = codecs[key]
    x = list(range(1, n + 1))
    kind = rnd.random()
    if kind < 0.15:
        msg = [0] * k
    elif kind < 0.45:
        keep = rnd.randrange(k + 1)
        msg = [rnd.randrange(p) for _ in range(keep)] + [0] * (k - keep)
    else:
        msg = [rnd.randrange(p) for _ in range(k)]
    word = [sum(c * pow(xi, e, p) for e, c in enumerate(msg)) % p for xi in x]
    cmax = n - 2 * t - 1
    nn = rnd.randrange(0, cmax + 2) if rnd.random() < 0.4 else 0          # (now and then one erasure too many: the assertion)
    nn = min(nn, n)
    for pos in rnd.sample(range(n), nn):
        word[pos] = None
    live = [i for i in range(n) if word[i] is not None]
    emax = max(0, (len(live) - t) // 2)
    ne = min(len(live), rnd.choice([0, emax, rnd.randrange(emax + 1), emax + 1, rnd.randrange(len(live) + 1)]))
    for pos in rnd.sample(live, ne):
        word[pos] = (word[pos] + rnd.randrange(1, p)) % p
    fpw = gg.GF(p)
    try:
        out = dec([None if w is None else fpw(w) for w in word], debug=False)
        want = ([c.value for c in out], None)
    except AssertionError:
        want = (None, "assert")
    except Exception as e:  # noqa: BLE001 - the reference raises bare Exceptions
        want = (None, str(e))
    co, st = oracle.wb_decode_batch(x, k, [word], p)[0]
    got = (co, None) if st == 0 else (None, oracle.WB_MESSAGES.get(st, "assert"))
    if want != got:
        fails += 1
        print("FAIL", p, n, k, "word", word, "reference", want, "oracle", got, flush=True)
    words += 1
    refused += want[0] is None
print(f"diff_wb_vs_reference: {words} words ({refused} refused by the reference), {fails} differences (seed {seed}, {budget:.0f} s)")
