"""Messages and error patterns that uniformly random draws never produce, for every decoder entry point's tests (and scratch/stress_gao.py).

Round 4's stress run found hb_wb_decode accepting Gao's result beyond the radius for messages with leading zeros -- what `chunk_data`'s
zero padding (reference utils/misc.py:33-51) makes of the last chunk of EVERY open -- after three rounds of green tests on uniformly random
coefficients.  These generators put such words into the suites themselves."""


def structured_message(rnd, k, p, kind=None):
    """one message of k coefficients: kind in {"zero", "constant", "short", "padded", "one_high", "full"} (None: drawn)"""
    if kind is None:
        kind = rnd.choice(["zero", "constant", "short", "padded", "one_high", "full", "full", "full"])
    if kind == "zero":
        return [0] * k
    if kind == "constant":
        return [rnd.randrange(p)] + [0] * (k - 1)
    if kind == "short":
        cut = rnd.randrange(k + 1)
        return [rnd.randrange(p) for _ in range(cut)] + [0] * (k - cut)
    if kind == "padded":                       # the last chunk of an open: a few shares, then chunk_data's zeros
        cut = rnd.randrange(1, max(2, k // 2 + 1))
        return [rnd.randrange(p) for _ in range(cut)] + [0] * (k - cut)
    if kind == "one_high":                     # only the leading coefficient
        return [0] * (k - 1) + [rnd.randrange(1, p) if p > 1 else 0]
    return [rnd.randrange(p) for _ in range(k)]


KINDS = ["zero", "constant", "short", "padded", "one_high"]


def structured_rows(rnd, p, c, d):
    """c rows of d coefficients: the structured kinds first (as many as fit in half the rows), uniform draws after them, and a padded last row"""
    rows = [[rnd.randrange(p) for _ in range(d)] for _ in range(c)]
    for i, kind in enumerate(KINDS):
        if i < c // 2:
            rows[i] = structured_message(rnd, d, p, kind)
    if c > 2 and d > 1:
        rows[-1] = structured_message(rnd, d, p, "padded")
    return rows


def coordinated_errors(rnd, word, x, k, ne, p, evaluate):
    """`ne` positions of the codeword replaced by the values of ANOTHER polynomial of degree < k (liars that agree with each other):
    -> (word, positions).  evaluate(x, coeffs) -> values at the points x."""
    other = [rnd.randrange(p) for _ in range(k)]
    vals = evaluate(x, other)
    word = list(word)
    pos = sorted(rnd.sample(range(len(word)), ne))
    changed = []
    for i in pos:
        if vals[i] != word[i]:
            word[i] = vals[i]
            changed.append(i)
    return word, changed
