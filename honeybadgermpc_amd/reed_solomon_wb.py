"""
Welch-Berlekamp Reed-Solomon decoder
(reference: honeybadgermpc/reed_solomon_wb.py:47-273, pure Python there).

`decode` -- the function WelchBerlekampRobustDecoder calls per codeword -- runs on the
GPU (hb_wb_decode: the (E, Q) linear system, Gauss-Jordan and the exact division, one
workgroup per codeword).  The linear-algebra helpers the reference module also exports
(`rref`, `no_solution`, `is_pivot_column`, `some_solution`, `solve_system`) are kept as
small host utilities with the reference's semantics.
"""
import logging

from .field import GF
from .polynomial import EvalPoint, polynomials_over


def make_wb_encoder_decoder(n, k, p, point=None):
    """n symbols per codeword, k = t+1 message symbols, prime p (reference :47-153).
    Returns (encode, decode, solve_system)."""
    if not k <= n <= p:
        raise Exception("Must have k <= n <= p but instead had (n,k,p) == (%r, %r, %r)" % (n, k, p))
    t = k - 1
    fp = GF(p)
    poly = polynomials_over(fp)
    if point is None or type(point) is not EvalPoint:
        point = EvalPoint(fp, n, use_omega_powers=False)
    xs = [point(i).value for i in range(n)]

    def encode(message):
        if not all(x < p for x in message):
            raise Exception("Message is improperly encoded as integers < p. It was:\n%r" % message)
        assert len(message) == t + 1
        from .ntl import vandermonde_batch_evaluate

        return [fp(v) for v in vandermonde_batch_evaluate(xs, [list(message)], p)[0]]

    def solve_system(encoded_message, max_e, debug=False):
        """(Q, E) for the largest e <= max_e whose system yields E | Q (reference :79-127).
        encoded_message: [(a_i, b_i)] field-element pairs."""
        for e in range(max_e, 0, -1):
            n_e, n_q = e + 1, e + k
            system = []
            for a, b in encoded_message:
                pw, row_e, row_q = fp(1), [], []
                for j in range(n_q):
                    if j < n_e:
                        row_e.append(b * pw)
                    row_q.append(-pw)
                    pw = pw * a
                system.append(row_e + row_q + [fp(0)])
            # force E monic of degree e
            system.append([fp(0)] * (n_e - 1) + [fp(1)] + [fp(0)] * n_q + [fp(1)])
            solution = some_solution(system, free_variable_value=1)
            e_ = poly(solution[:n_e])
            q_ = poly(solution[n_e:])
            if debug:
                logging.debug("e=%r Q=%r E=%r", e, q_, e_)
            _, remainder = divmod(q_, e_)
            if remainder.is_zero():
                return q_, e_
        raise ValueError("found no divisors!")

    def decode(encoded_msg, debug=True):
        """encoded_msg: n entries, GFElement / int, or None for an erasure.  Returns the
        message polynomial's coefficient list with trailing zeros stripped (so it may be
        shorter than k, and [] for the zero message; reference tests :68,75)."""
        assert len(encoded_msg) == n
        c = sum(m is None for m in encoded_msg)
        assert 2 * t + 1 + c <= n
        if debug:
            logging.debug(f"n: {n} k: {k} t: {t} c: {c}")
        from .device import wb_decode_batch

        row = [None if m is None else int(m) for m in encoded_msg]
        coeffs, status = wb_decode_batch(xs, k, [row], p)[0]
        if status == 1:
            raise ValueError("found no divisors!")  # reference :127
        if status == 2:
            raise Exception("No solution")  # reference :245
        if status != 0:
            raise Exception(f"wb_decode failed with status {status}")
        return [fp(v) for v in coeffs]

    return encode, decode, solve_system


# ---------------------------------------------------------------------------
# host linear-algebra helpers with the reference's semantics (:157-273)
# ---------------------------------------------------------------------------
def rref(matrix):
    """In-place reduced row echelon form; pivot = first non-zero entry at or below the
    current row (reference :157-197)."""
    if not matrix:
        return
    n_rows, n_cols = len(matrix), len(matrix[0])
    i = j = 0
    while i < n_rows and j < n_cols:
        if matrix[i][j] == 0:
            r = i
            while r < n_rows and matrix[r][j] == 0:
                r += 1
            if r == n_rows:
                j += 1
                continue
            matrix[i], matrix[r] = matrix[r], matrix[i]
        pivot = matrix[i][j]
        matrix[i] = [x / pivot for x in matrix[i]]
        for r in range(n_rows):
            if r != i and matrix[r][j] != 0:
                f = matrix[r][j]
                matrix[r] = [y - f * x for x, y in zip(matrix[i], matrix[r])]
        i += 1
        j += 1
    return matrix


def no_solution(a):
    """(True, 0) when the last non-zero row of a reduced system reads 0 = c (reference :203-214)."""
    i = -1
    while all(x == 0 for x in a[i]):
        i -= 1
    if all(x == 0 for x in a[i][:-1]):
        return True, 0
    return False, i


def is_pivot_column(a, j):
    """(is_pivot, row): column j is all zeros except a single 1 (reference :217-237)."""
    i = 0
    while i < len(a) and a[i][j] == 0:
        i += 1
    if i == len(a):
        return (False, i)
    if a[i][j] != 1:
        return (False, i)
    pivot_row = i
    for r in range(pivot_row + 1, len(a)):
        if a[r][j] != 0:
            return (False, pivot_row)
    return (True, pivot_row)


def some_solution(system, free_variable_value=1):
    """One solution of the augmented system, free variables set to the given value
    (reference :240-273)."""
    rref(system)
    if no_solution(system)[0]:
        raise Exception("No solution")
    num_vars = len(system[0]) - 1
    values = [0] * num_vars
    pivot_row_of = {}
    for j in range(num_vars):
        is_pivot, row = is_pivot_column(system, j)
        if is_pivot:
            pivot_row_of[j] = row
    free = [j for j in range(num_vars) if j not in pivot_row_of]
    for j in free:
        values[j] = free_variable_value
    for j, row in pivot_row_of.items():
        values[j] = system[row][-1] - sum(system[row][i] * values[i] for i in free)
    return values
