"""
Randomness extraction through the codec
(reference: honeybadgermpc/progs/random_refinement.py:5-19).

k contributed random shares are read as the coefficients of a polynomial, which is evaluated at
the omega powers; dropping t outputs removes whatever t corrupt contributors could bias.  One
encoder call, so it runs on the same kernels as the R1 encode.
"""
from ..polynomial import EvalPoint
from ..reed_solomon import EncoderFactory


def refine_randoms(n, t, field, random_shares_int):
    assert 3 * t + 1 <= n
    k = len(random_shares_int)          # contributors to this batch
    assert k >= n - t and k <= n
    encoder = EncoderFactory.get(EvalPoint(field, n, use_omega_powers=True))
    output_shares_int = encoder.encode(random_shares_int)
    return output_shares_int[: k - t]
