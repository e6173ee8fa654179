"""The candidate rule of the device decoder's robust phase (device.py _candidate_cap / _track_candidates) against the reference's own
loop on the CPU.  The reference (reed_solomon.py:334-346), for ONE polynomial and after every arrival: run Gao over the arrived symbols; on
(Q, errors) accept when |z| - |errors| >= need = degree + 1 + max_errors - confirmed, otherwise wait with nothing changed.  The rule: a
polynomial P of degree <= `degree` that disagrees with E of the arrived senders, E <= max_errors - confirmed, is the only polynomial the
reference can accept; "accept P with exactly its E senders when |z| - E >= need, else wait" is then what the reference does.  Checked
prefix by prefix with the oracle's Gao (rsdecode_impl.h:281-363) as the reference's decoder: honest words, garbage, liars coordinated on
another polynomial that shares points with the true one, more liars than max_errors, confirmed errors already removed."""
import random

import oracle
from conftest import BLS
from structured import structured_message


def ev(f, x, p):
    acc = 0
    for c in reversed(f):
        acc = (acc * x + c) % p
    return acc


def reference_step(xs, ys, k, p, need):
    """what the reference does with the arrivals so far: None (wait) or (coefficients, indices in error)"""
    co, el = oracle.gao_interpolate(xs, ys, k, p)
    if co is None:
        return None
    errors = [j for j, x in enumerate(xs) if ev(el, x, p) == 0] if len(el) > 1 else []
    if len(xs) - len(errors) < need:
        return None
    return co, errors


def test_candidate_rule_is_the_reference_loop():
    rnd = random.Random(9)
    prefixes = accepted = governed = fake_accepted = dropped = 0
    for trial in range(1200):
        p = rnd.choice([BLS, BLS, 257, 10007])
        t = rnd.randrange(1, 8)
        n = rnd.randrange(3 * t + 1, 3 * t + 6)
        degree, k = t, t + 1
        confirmed = rnd.randrange(0, t) if rnd.random() < 0.3 else 0          # senders expelled by earlier polynomials of the batch
        need = degree + 1 + t - confirmed
        cap = t - confirmed
        n_av = n - confirmed                                                     # those senders never count again
        xs_all = rnd.sample(range(1, min(p, 10 ** 6)), n_av)
        # (a third of the true polynomials are the structured ones uniform draws never produce -- zero, constant, short, padded: Gao decodes
        # those BEYOND the radius of a full-degree message, which is where the rule's argument has to hold too)
        true = structured_message(rnd, k, p) if trial % 3 == 0 else [rnd.randrange(p) for _ in range(k)]
        # the liars' word: garbage, or one fake polynomial equal to the true one at `shared` points
        n_liars = min(n_av, rnd.choice([0, 1, cap, cap, cap, cap + 1, cap + 2, rnd.randrange(0, n_av + 1)]))
        liars = set(rnd.sample(range(n_av), n_liars))
        fake = None
        if rnd.random() < 0.6:
            q, deg = [rnd.randrange(1, p)] + [0] * degree, 0
            for s in rnd.sample(range(n_av), rnd.randrange(0, degree + 1)):
                nq = [0] * k
                for e in range(deg + 1):
                    nq[e + 1] = (nq[e + 1] + q[e]) % p
                    nq[e] = (nq[e] - q[e] * xs_all[s]) % p
                q, deg = nq, deg + 1
            fake = [(a + b) % p for a, b in zip(true, q)]
        word = [(ev(fake, xs_all[j], p) if fake is not None else rnd.randrange(p)) if j in liars else ev(true, xs_all[j], p) for j in range(n_av)]
        order = list(range(n_av))
        rnd.shuffle(order)
        if rnd.random() < 0.5:
            order = sorted(liars) + [j for j in order if j not in liars]
        # the candidates the device decoder would hold: interpolants of the oldest / newest degree + 1 arrivals at the moment `need`
        # columns are in -- here simply the two polynomials in play (a contaminated interpolant is "some other polynomial": the fake
        # stands in for it), each tracked while its disagreements stay within the cap
        cands = [c for c in (true, fake) if c is not None]
        for m in range(k, n_av + 1):
            xs, ys = [xs_all[j] for j in order[:m]], [word[j] for j in order[:m]]
            want = reference_step(xs, ys, k, p, need)
            prefixes += 1
            for P in cands:
                E = [j for j in range(m) if ev(P, xs[j], p) != ys[j]]
                if len(E) > cap:
                    dropped += 1
                    continue                                                     # the rule says nothing: the decoder falls back to the probe
                governed += 1
                if m - len(E) >= need:
                    assert want is not None and want[0] == P + [0] * (k - len(P)) and sorted(want[1]) == E, (trial, m, "rule accepts, reference differs")
                    accepted += 1
                    fake_accepted += P is fake
                else:
                    assert want is None, (trial, m, "rule waits, the reference accepted", want)
            if want is not None:
                break                                                            # the polynomial is settled; the reference moves on
    assert prefixes > 3000 and governed > 2500 and accepted > 300 and fake_accepted > 10 and dropped > 200, (prefixes, governed, accepted, fake_accepted, dropped)


def test_no_acceptance_before_the_bound_a_verdict_implies():
    """The rule that lets the decoder skip probe verdicts (device.py, the stalled branch): Gao failing over m columns, or decoding with e
    errors short of support, implies the reference accepts nothing before need + ((m - degree - 1) // 2 + 1), respectively need + e,
    columns are in.  Every prefix of random words (garbage liars, coordinated liars, too many liars), the oracle's Gao as the decoder."""
    rnd = random.Random(21)
    implied = checked = accepted = 0
    for trial in range(750):
        p = rnd.choice([BLS, 257, 10007])
        t = rnd.randrange(1, 8)
        n = rnd.randrange(3 * t + 1, 3 * t + 6)
        k = t + 1
        need = k + t
        xs_all = rnd.sample(range(1, min(p, 10 ** 6)), n)
        true = structured_message(rnd, k, p) if trial % 3 == 0 else [rnd.randrange(p) for _ in range(k)]
        liars = set(rnd.sample(range(n), min(n, rnd.choice([1, t, t, t, t + 1, rnd.randrange(0, n + 1)]))))
        fake = [rnd.randrange(p) for _ in range(k)] if rnd.random() < 0.5 else None
        word = [(ev(fake, xs_all[j], p) if fake is not None else rnd.randrange(p)) if j in liars else ev(true, xs_all[j], p) for j in range(n)]
        order = list(range(n))
        rnd.shuffle(order)
        earliest = 0                                   # no acceptance may happen with fewer columns than this
        for m in range(k, n + 1):
            xs, ys = [xs_all[j] for j in order[:m]], [word[j] for j in order[:m]]
            co, el = oracle.gao_interpolate(xs, ys, k, p)
            errors = None if co is None else ([j for j, x in enumerate(xs) if ev(el, x, p) == 0] if len(el) > 1 else [])
            ok = errors is not None and m - len(errors) >= need
            checked += 1
            if ok:
                assert m >= earliest, (trial, m, earliest)
                accepted += 1
                break
            bound = need + ((m - k) // 2 + 1 if errors is None else len(errors))
            if bound > m + 1:
                implied += 1
            earliest = max(earliest, bound)
    assert checked > 2000 and implied > 300 and accepted > 100, (checked, implied, accepted)
