"""The decision rule of the device probe (hb_quick.hip) against the oracle's Gao (reference rsdecode_impl.h:281-363) on the CPU:
every prefix of random arrival orders, inside and beyond the unique-decoding radius, several fields."""
import random

import oracle
from conftest import BLS
from probe_rule import Probe, ev


def test_probe_rule_equals_gao_on_every_prefix():
    rnd = random.Random(5)
    trials = decoded = beyond = 0
    for _ in range(700):
        p = rnd.choice([BLS, 53, 13, 257])
        n = rnd.randrange(2, 32 if p == BLS else min(p - 1, 30))
        k = rnd.randrange(1, n + 1)
        xs_all = rnd.sample(range(1, min(p, 10 ** 6)), n)
        arrive = list(range(n))
        rnd.shuffle(arrive)
        arrive = arrive[: rnd.randrange(1, n + 1)]
        n1 = len(arrive)
        f = [rnd.randrange(p) for _ in range(k)]
        e = max((n1 - k) // 2, 0)
        nerr = min(rnd.choice([0, 0, 1, e, e, e + 1, e + 1, e + 2, rnd.randrange(0, n1 + 1)]), n1)
        bad = set(rnd.sample(range(n1), nerr))
        xs = [xs_all[i] for i in arrive]
        ys = [rnd.randrange(p) if j in bad else ev(f, x, p) for j, x in enumerate(xs)]
        pr = Probe(k, p, n1 + 1)
        for step, (x, y) in enumerate(zip(xs, ys)):
            pr.add(x, y)
            m = step + 1
            if m < k:
                continue
            got = pr.decide(xs_all)
            co, el = oracle.gao_interpolate(xs[:m], ys[:m], k, p)
            want = None
            if co is not None:
                want = [i for i, x in enumerate(xs_all) if ev(el, x, p) == 0] if len(el) > 1 else []
                beyond += sum(1 for j in range(m) if ev(co, xs[j], p) != ys[j]) > (m - k) // 2
            trials += 1
            assert (got is None) == (want is None) and (got is None or sorted(got) == sorted(want)), (p, n, k, m, got, want)
            decoded += want is not None
    assert trials > 1500 and decoded > 500 and beyond > 5, (trials, decoded, beyond)
