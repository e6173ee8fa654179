"""
Randomness extraction through the codec (replaces honeybadgermpc/progs/random_refinement.py:5-19).

The k contributed shares are the coefficients of a polynomial; its values at the first k - t omega points are the refined
shares (t fewer than were contributed: that many contributors may be corrupt and could bias as many outputs).  It is one
encode on the kernels of an open's R1 encode: `offline.extract_at_omega_powers` does it for lists of ints and for device
tensors of many batches alike.
"""
from ..offline import extract_at_omega_powers


def refine_randoms(n, t, field, random_shares_int):
    return extract_at_omega_powers(field, n, t, [random_shares_int])[0]
