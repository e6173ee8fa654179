#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (runs only where /root/reference exists).

Randomised differential run of this repo's HOST batch_reconstruct coroutine (honeybadgermpc_amd/batch_reconstruction.py, arithmetic on the
oracle) against the REFERENCE's own batch_reconstruct (honeybadgermpc/batch_reconstruction.py:88-227) over the reference's SimpleRouter:
for the same share vectors -- up to t parties holding garbage, batch sizes that leave a padded last chunk, secrets with special values,
both point policies, Gao and Welch-Berlekamp -- every party's output and every R1 / R2 message every party sends must be equal.

    python oracle/diff_batch_reconstruct_vs_reference.py [seconds] [seed]
"""
import asyncio
import logging
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

import oracle  # noqa: E402
import honeybadgermpc_amd.device as dev  # noqa: E402
import honeybadgermpc_amd.ntl as ntl  # noqa: E402
import honeybadgermpc_amd.polynomial as poly  # noqa: E402
import honeybadgermpc_amd.reed_solomon as rs  # noqa: E402
from honeybadgermpc_amd.batch_reconstruction import batch_reconstruct as our_batch_reconstruct  # noqa: E402
from honeybadgermpc_amd.field import GF as OurGF  # noqa: E402

for name in ("lagrange_interpolate", "evaluate", "vandermonde_batch_interpolate", "vandermonde_batch_evaluate", "fft", "partial_fft",
             "fft_batch_evaluate", "fft_interpolate", "fft_batch_interpolate", "gao_interpolate", "gao_interpolate_batch",
             "vandermonde_inverse", "sqrt_mod"):
    setattr(ntl, name, getattr(oracle, name))
    if hasattr(rs, name):
        setattr(rs, name, getattr(oracle, name))
poly.fft_cpp = oracle.fft
poly.fft_interpolate_cpp = oracle.fft_interpolate
dev.wb_decode_batch = oracle.wb_decode_batch
ntl.InterpolationError = oracle.InterpolationError
logging.disable(logging.CRITICAL)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
BLS = gg.BLS


def run(impl, field, p, t, n, shares, use_omega, robust):
    sent = [{"R1": [None] * n, "R2": None} for _ in range(n)]

    async def go():
        router = gg.SimpleRouter(n)
        tasks = []
        for i in range(n):
            def mk(i):
                base = router.sends[i]

                def send(dest, msg):
                    tag, payload = msg
                    if tag == "R1":
                        sent[i]["R1"][dest] = [int(v) for v in payload]
                    else:
                        sent[i]["R2"] = [int(v) for v in payload]
                    base(dest, msg)
                return send
            tasks.append(impl([field(v) for v in shares[i]], p, t, n, i, mk(i), router.recvs[i], config=gg._Cfg(robust), use_omega_powers=use_omega))
        futs = [asyncio.ensure_future(tk) for tk in tasks]
        # a party whose decoder raises (Welch-Berlekamp's "No solution" propagates, reed_solomon.py:205-212) leaves the others waiting
        # for its R2 message: the reference hangs there, and so must the mirror -- parties still waiting after 3 s are recorded as such
        await asyncio.wait(futs, timeout=3)
        outs = []
        for f in futs:
            if not f.done():
                f.cancel()
                outs.append("waiting")
            elif f.exception() is not None:
                outs.append(("exception", type(f.exception()).__name__, str(f.exception())))
            else:
                r = f.result()
                outs.append(None if r is None else [int(v.value) for v in r])
        await asyncio.gather(*futs, return_exceptions=True)
        return outs

    return asyncio.run(go()), sent


t_end = time.time() + budget
runs = fails = order_dependent = 0
while time.time() < t_end:
    p = BLS
    n = rnd.choice([4, 7, 10, 13])
    t = rnd.randrange(1, (n - 1) // 3 + 1)
    b = rnd.choice([1, t, t + 1, t + 2, 2 * (t + 1) + 1, rnd.randrange(1, 30)])
    use_omega = rnd.random() < 0.3
    robust = rnd.choice(["gao", "gao", "welch-berlekamp"])
    fp = gg.GF(p)
    point = gg.EvalPoint(fp, n, use_omega_powers=use_omega)
    xs = [point(i).value for i in range(n)]
    secrets = [rnd.choice([0, 1, p - 1]) if rnd.random() < 0.25 else rnd.randrange(p) for _ in range(b)]
    polys = [[s] + [rnd.choice([0, rnd.randrange(p)]) for _ in range(t)] for s in secrets]
    shares = [[sum(c * pow(xs[i], e, p) for e, c in enumerate(pl)) % p for pl in polys] for i in range(n)]
    bad = rnd.sample(range(n), rnd.randrange(0, t + 1))
    for i in bad:
        kind = rnd.randrange(3)
        if kind == 0:
            shares[i] = [rnd.randrange(p) for _ in range(b)]
        elif kind == 1:
            shares[i] = [(v + 1) % p for v in shares[i]]
        else:
            shares[i] = list(shares[i])
            shares[i][-1] = (shares[i][-1] + rnd.randrange(1, p)) % p
    want = run(gg.batch_reconstruct, fp, p, t, n, shares, use_omega, robust)
    got = run(our_batch_reconstruct, OurGF(p), p, t, n, shares, use_omega, robust)
    # Where some party of the REFERENCE run raises (Welch-Berlekamp's "No solution" beyond the radius; an IndexError of the reference's
    # own when its stripped rows meet flatten / truncation), WHICH parties do depends on the order in which each party happens to
    # drain its queue -- the scheduling of two different coroutine bodies, not the decoders (those are held to the reference column by
    # column, on identical arrival orders, by oracle/diff_incremental_vs_reference.py).  Such runs are compared on their R1 messages,
    # which no arrival order influences, and counted apart.
    fragile = any(not isinstance(o, list) for o in want[0] + got[0]) and robust != "gao"        # (either run: the same inputs in another order)
    if fragile:
        order_dependent += 1
        same = all(want[1][i]["R1"] == got[1][i]["R1"] for i in range(n))
    else:
        same = want == got
    if not same:
        fails += 1
        def short(o):
            return o if not isinstance(o, list) else ("secrets" if o == secrets else f"list of {len(o)} != secrets")
        print("FAIL", n, t, b, use_omega, robust, "bad", bad, "secrets", secrets if b <= 4 else f"({b})", "polys", polys if b <= 4 else "", "shares", shares if b <= 4 else "",
              "\n   reference:", [short(o) for o in want[0]], "\n   ours:     ", [short(o) for o in got[0]],
              "\n   R1 equal:", [want[1][i]["R1"] == got[1][i]["R1"] for i in range(n)], "R2 equal:", [want[1][i]["R2"] == got[1][i]["R2"] for i in range(n)], flush=True)
    runs += 1
print(f"diff_batch_reconstruct_vs_reference: {runs} opens (every party's output and every message sent; {order_dependent} of them order-dependent in the "
      f"reference itself: R1 messages only), {fails} differences (seed {seed}, {budget:.0f} s)")
