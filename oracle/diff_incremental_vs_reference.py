#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (runs only where /root/reference exists; nothing of the product imports it).

Randomised differential run of this repo's HOST IncrementalDecoder (honeybadgermpc_amd/reed_solomon.py, arithmetic routed to the CPU
oracle as tests/conftest.py does) against the REFERENCE's own IncrementalDecoder (honeybadgermpc/reed_solomon.py:232-403, imported from
/root/reference the way oracle/gen_golden.py imports it, its NTL calls served by the oracle): after EVERY column the two must agree on
done(), the results (row by row, Welch-Berlekamp's stripped rows included), the confirmed errors, and raise the same exception at the
same column.  Inputs are structured as well as random: polynomials with leading zeros (chunk_data's padding), zero polynomials, liars
that are random / coordinated on one other polynomial / one too many, arrival orders with the liars first.

    python oracle/diff_incremental_vs_reference.py [seconds] [seed]
"""
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the reference package with the oracle behind honeybadgermpc.ntl)

import oracle  # noqa: E402
import honeybadgermpc_amd.device as dev  # noqa: E402
import honeybadgermpc_amd.ntl as ntl  # noqa: E402
import honeybadgermpc_amd.polynomial as poly  # noqa: E402
import honeybadgermpc_amd.reed_solomon as rs  # noqa: E402
from honeybadgermpc_amd.field import GF as OurGF  # noqa: E402
from honeybadgermpc_amd.polynomial import EvalPoint as OurEvalPoint  # noqa: E402

# the package's arithmetic entry points -> the oracle (tests/conftest.py: install_oracle_backend)
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

import logging  # noqa: E402

logging.disable(logging.CRITICAL)
ref_rs = gg.ref_rs
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
BLS = gg.BLS
t_end = time.time() + budget
runs = fails = raised = 0
while time.time() < t_end:
    p = rnd.choice([BLS, BLS, 53, 257])
    n = rnd.choice([4, 7, 10, 13, 16])
    if p == 53 and n > 13:
        n = 13
    t = rnd.randrange(1, (n - 1) // 3 + 1)
    c = rnd.choice([1, 2, 3, 5])
    use_omega = p == BLS and rnd.random() < 0.3
    robust = rnd.choice(["gao", "gao", "welch-berlekamp"])
    rpoint = gg.EvalPoint(gg.GF(p), n, use_omega_powers=use_omega)
    opoint = OurEvalPoint(OurGF(p), n, use_omega_powers=use_omega)
    xs = [rpoint(i).value for i in range(n)]
    polys = [[rnd.randrange(p) for _ in range(t + 1)] for _ in range(c)]
    if rnd.random() < 0.5:
        for j in ([c - 1] if rnd.random() < 0.5 else range(c)):
            keep = rnd.randrange(0, t + 1)
            polys[j] = polys[j][:keep] + [0] * (t + 1 - keep)
    cols = [[sum(co * pow(xs[i], e, p) for e, co in enumerate(pl)) % p for pl in polys] for i in range(n)]
    liars = rnd.sample(range(n), min(n, rnd.randrange(0, t + 2)))
    fake = [[rnd.randrange(p) for _ in range(t + 1)] for _ in range(c)] if rnd.random() < 0.5 else None
    for i in liars:
        for j in (range(c) if rnd.random() < 0.6 else [rnd.randrange(c)]):
            if fake is not None:
                cols[i][j] = sum(co * pow(xs[i], e, p) for e, co in enumerate(fake[j])) % p
            else:
                cols[i][j] = (cols[i][j] + rnd.randrange(1, p)) % p
    order = list(range(n))
    rnd.shuffle(order)
    if rnd.random() < 0.5:
        order = liars + [i for i in order if i not in liars]
    ralgo = ref_rs.Algorithm.FFT if use_omega else ref_rs.Algorithm.VANDERMONDE
    oalgo = rs.Algorithm.FFT if use_omega else rs.Algorithm.VANDERMONDE
    ref = ref_rs.IncrementalDecoder(ref_rs.EncoderFactory.get(rpoint, ralgo), ref_rs.DecoderFactory.get(rpoint, ralgo),
                                    ref_rs.RobustDecoderFactory.get(t, rpoint, algorithm=robust), degree=t, batch_size=c, max_errors=t)
    ours = rs.IncrementalDecoder(rs.EncoderFactory.get(opoint, oalgo), rs.DecoderFactory.get(opoint, oalgo),
                                 rs.RobustDecoderFactory.get(t, opoint, algorithm=robust), degree=t, batch_size=c, max_errors=t)
    bad = None
    for step, idx in enumerate(order):
        rexc = oexc = None
        try:
            ref.add(idx, list(cols[idx]))
        except BaseException as e:  # noqa: BLE001 - the reference raises bare Exceptions and AssertionErrors
            rexc = e
        try:
            ours.add(idx, list(cols[idx]))
        except BaseException as e:  # noqa: BLE001
            oexc = e
        if (rexc is None) != (oexc is None) or (rexc is not None and (type(rexc) is not type(oexc) or str(rexc) != str(oexc))):
            bad = ("exception", step, repr(rexc), repr(oexc))
            break
        if rexc is not None:
            raised += 1
            break
        rres, rerr = ref.get_results()
        ores, oerr = ours.get_results()
        if ref.done() != ours.done() or (rres is None) != (ores is None):
            bad = ("done", step, ref.done(), ours.done())
            break
        if rres is not None:
            if [list(map(int, r)) for r in rres] != [list(map(int, r)) for r in ores] or set(rerr) != set(oerr):
                bad = ("result", step, rres, ores, sorted(rerr), sorted(oerr))
            break
    if bad:
        fails += 1
        print("FAIL", p, n, t, c, use_omega, robust, "polys", polys, "liars", liars, "order", order, bad, flush=True)
    runs += 1
print(f"diff_incremental_vs_reference: {runs} decodes against the reference's own IncrementalDecoder ({raised} ended in the reference's exception), "
      f"{fails} differences (seed {seed}, {budget:.0f} s)")
