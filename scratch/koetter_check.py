"""Design check (CPU, pure Python ints) for the single-codeword probe of the device decoder: Koetter / Welch-Berlekamp rational
interpolation, one point at a time, fraction-free.  Decision rule under test (no division, no inversion):
    Q = A + B Y of minimal (1, k-1)-weighted degree through all points decodes  <=>
    B != 0, deg A <= deg B + k - 1, deg B <= e = (n' - k) // 2 and B has deg B distinct roots among the points
and then the error positions are those roots.  Compared with the oracle's Gao (reference rsdecode_impl.h:325-363) on random
words inside and beyond the radius."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from probe_rule import Probe, ev  # noqa: E402


def main():
    rnd = random.Random(5)
    trials = agree = decoded = beyond = 0
    for it in range(4000):
        p = rnd.choice([P, 53, 13, 257])
        n = rnd.randrange(2, 40 if p == P else min(p - 1, 30))
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
                wrong = sum(1 for j in range(m) if ev(co, xs[j], p) != ys[j])
                beyond += wrong > (m - k) // 2
            trials += 1
            if (got is None) == (want is None) and (got is None or sorted(got) == sorted(want)):
                agree += 1
            else:
                print("MISMATCH", it, p, n, k, m, got, want)
                return 1
            decoded += want is not None
    print(f"{agree}/{trials} prefix decisions agree with the oracle's Gao ({decoded} decodable, {beyond} of them beyond the unique-decoding radius)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
