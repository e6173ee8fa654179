"""
Triple extraction through the codec
(reference: honeybadgermpc/progs/triple_refinement.py:8-88).

m "dirty" multiplication triples, shared with degree t, are read as points 0 .. m-1 of polynomials
A, B (degree d = (m-1)//2) and C = A B (degree 2d).  A and B are fixed by the first d+1 triples; the
d missing points of C come from one batched Beaver multiplication that spends the other triples; C is
then interpolated from 2d+1 points and all three are evaluated at k = d + 1 - t fresh points.  Every
interpolation / evaluation is a Vandermonde call of the ntl drop-in, i.e. it runs on the same kernels
as the batch open.

`context` is whatever the caller's MPC runtime is (out of scope here): it needs N, t, field and
ShareArray(list_of_ints) with `-` and an awaitable `.open()` returning field elements.
"""
import asyncio

from ..ntl import vandermonde_batch_evaluate, vandermonde_batch_interpolate


async def batch_beaver(context, a_, b_, x_, y_, z_):
    """Shares of a_i * b_i, spending the triples (x_i, y_i, z_i) (reference :8-18)."""
    assert len(a_) == len(b_) == len(x_) == len(y_) == len(z_)
    a, b, x, y = (context.ShareArray(v) for v in (a_, b_, x_, y_))
    f, g = await asyncio.gather((a - x).open(), (b - y).open())
    # (a - x)(b - y) + (a - x) y + (b - y) x + x y, the first factor public
    return [(d * e).value + (d * q).value + (e * p).value + pq for (p, q, pq, d, e) in zip(x_, y_, z_, f, g)]


async def refine_triples(context, a_dirty, b_dirty, c_dirty):
    """-> (p, q, pq): k = d + 1 - t refined triples' shares as lists of ints (reference :21-88)."""
    assert len(a_dirty) == len(b_dirty) == len(c_dirty)
    n, t = context.N, context.t
    m = len(a_dirty)
    d = (m - 1) // 2
    modulus = context.field.modulus
    assert n - t <= m <= n

    def interpolate(points, values):
        return vandermonde_batch_interpolate(points, [values], modulus)[0]

    def evaluate(points, coeffs):
        return vandermonde_batch_evaluate(points, [coeffs], modulus)[0]

    first = list(range(d + 1))
    a_coeffs, b_coeffs = interpolate(first, a_dirty[: d + 1]), interpolate(first, b_dirty[: d + 1])
    assert len(a_coeffs) == len(b_coeffs) == d + 1

    rest = list(range(d + 1, 2 * d + 1))
    a_rest, b_rest = evaluate(rest, a_coeffs), evaluate(rest, b_coeffs)
    assert len(a_rest) == len(b_rest) == d

    spend = slice(d + 1, 2 * d + 1)
    c_rest = await batch_beaver(context, a_rest, b_rest, a_dirty[spend], b_dirty[spend], c_dirty[spend])
    assert len(c_rest) == d

    c_coeffs = interpolate(list(range(2 * d + 1)), c_dirty[: d + 1] + c_rest)
    assert len(c_coeffs) == 2 * d + 1

    k = d + 1 - t                       # triples that can be extracted securely
    fresh = list(range(n + 1, n + 1 + k))
    p, q, pq = evaluate(fresh, a_coeffs), evaluate(fresh, b_coeffs), evaluate(fresh, c_coeffs)
    assert len(p) == len(q) == len(pq) == k
    return p, q, pq
