"""
Triple extraction through the codec (replaces honeybadgermpc/progs/triple_refinement.py:8-88).

m "dirty" multiplication triples, shared with degree t, are read as points 0 .. m-1 of polynomials A, B (degree
d = (m-1)//2) and C = A B (degree 2d).  A and B are fixed by the first d+1 triples; the d missing points of C come from one
batched Beaver multiplication that spends the other triples; C is then interpolated from 2d+1 points and all three are
evaluated at k = d + 1 - t fresh points.  A and B always travel together as a batch of two rows, and the last step evaluates
A, B and C in ONE call (the batched evaluate pads short rows with zeros, hbmpc_ntl_helpers.pyx:217,232-233): four calls of
the ntl drop-in where the reference makes nine, each on the kernels of the batch open.

`context` is whatever the caller's MPC runtime is (out of scope here): it needs N, t, field and ShareArray(list_of_ints)
with `-` and an awaitable `.open()` returning field elements.
"""
import asyncio

from ..ntl import vandermonde_batch_evaluate, vandermonde_batch_interpolate


async def batch_beaver(context, a_, b_, x_, y_, z_):
    """Shares of a_i b_i, spending the triples (x_i, y_i, z_i) (reference :8-18): with alpha = a - x and beta = b - y opened,
    a b = alpha beta + alpha y + beta x + x y."""
    count = len(z_)
    if any(len(v) != count for v in (a_, b_, x_, y_)):
        raise AssertionError("batch_beaver: vectors of different lengths")
    masked = [context.ShareArray(v) - context.ShareArray(mask) for v, mask in ((a_, x_), (b_, y_))]
    alpha, beta = await asyncio.gather(*(m.open() for m in masked))
    products = []
    for i in range(count):
        public = alpha[i] * beta[i]
        products.append(public.value + (alpha[i] * y_[i]).value + (beta[i] * x_[i]).value + z_[i])
    return products


async def refine_triples(context, a_dirty, b_dirty, c_dirty):
    """-> (p, q, pq): k = d + 1 - t refined triples' shares as lists of ints (reference :21-88)."""
    n, t, modulus = context.N, context.t, context.field.modulus
    m = len(a_dirty)
    if not (len(b_dirty) == len(c_dirty) == m and n - t <= m <= n):
        raise AssertionError(f"refine_triples: {m} triples with n = {n}, t = {t}")
    d = (m - 1) // 2
    defining, spent = slice(0, d + 1), slice(d + 1, 2 * d + 1)

    # A, B from their first d + 1 points; their values at the d points after those
    ab_coeffs = vandermonde_batch_interpolate(list(range(d + 1)), [a_dirty[defining], b_dirty[defining]], modulus)
    a_rest, b_rest = vandermonde_batch_evaluate(list(range(d + 1, 2 * d + 1)), ab_coeffs, modulus)

    # C = A B there, by Beaver multiplication with the triples of those positions
    c_rest = await batch_beaver(context, a_rest, b_rest, a_dirty[spent], b_dirty[spent], c_dirty[spent])
    c_points = c_dirty[defining] + c_rest
    (c_coeffs,) = vandermonde_batch_interpolate(list(range(len(c_points))), [c_points], modulus)

    k = d + 1 - t                                   # triples that can be extracted securely
    p, q, pq = vandermonde_batch_evaluate(list(range(n + 1, n + 1 + k)), [*ab_coeffs, c_coeffs], modulus)
    return p, q, pq
