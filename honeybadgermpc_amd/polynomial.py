"""
Polynomials over GF(p), roots of unity and the evaluation-point policy
(reference: honeybadgermpc/polynomial.py:14-423).

The Polynomial class is host-side Python on GFElement objects: it is the small-object API
(and the independent check the reference's tests use), not the bulk path.  Bulk evaluation
/ interpolation goes through honeybadgermpc_amd.ntl -> HIP kernels.
"""
import operator
from functools import reduce
from itertools import zip_longest

from .field import GF, GFElement
from .ntl import fft as fft_cpp
from .ntl import fft_interpolate as fft_interpolate_cpp


def strip_trailing_zeros(a):
    """Drop high-order zero coefficients; [] for the zero polynomial (reference :14-20)."""
    n = len(a)
    while n > 0 and a[n - 1] == 0:
        n -= 1
    return a[:n] if n else []


_poly_cache = {}


def polynomials_over(field):
    """Polynomial class bound to `field` (one class per field, reference :26-250).
    The reference also accepts the Rust pairing scalar type ZR; that crate is outside
    this path, so only GF fields are accepted here."""
    assert type(field) is GF
    if field in _poly_cache:
        return _poly_cache[field]

    class Polynomial(object):
        def __init__(self, coeffs):
            self.coeffs = list(strip_trailing_zeros(coeffs))
            for i, c in enumerate(self.coeffs):
                if type(c) is int:
                    self.coeffs[i] = field(c)
                assert type(self.coeffs[i]) is GFElement
            self.field = field

        # -- basics ----------------------------------------------------------
        def is_zero(self):
            return self.coeffs == [] or (len(self.coeffs) == 1 and self.coeffs[0] == 0)

        def __repr__(self):
            if self.is_zero():
                return "0"
            return " + ".join(f"{a} x^{i}" if i > 0 else f"{a}" for i, a in enumerate(self.coeffs))

        def __call__(self, x):
            acc, power = field(0), field(1)
            for c in self.coeffs:
                acc += c * power
                power *= x
            return acc

        def __eq__(self, other):
            return type(other) is Polynomial and other.coeffs == self.coeffs

        def __abs__(self):  # 1 + degree; 0 only for the zero polynomial
            return len(self.coeffs)

        def __iter__(self):
            return iter(self.coeffs)

        def __len__(self):
            return len(self.coeffs)

        def degree(self):
            return abs(self) - 1

        def leading_coefficient(self):
            return self.coeffs[-1]

        # -- arithmetic ------------------------------------------------------
        def __neg__(self):
            return Polynomial([-a for a in self])

        def __add__(self, other):
            return Polynomial([a + b for a, b in zip_longest(self, other, fillvalue=self.field(0))])

        def __sub__(self, other):
            return self + (-other)

        def __mul__(self, other):
            if self.is_zero() or other.is_zero():
                return Polynomial([])
            out = [self.field(0)] * (len(self) + len(other) - 1)
            for i, a in enumerate(self):
                for j, b in enumerate(other):
                    out[i + j] = out[i + j] + a * b
            return Polynomial(out)

        def __divmod__(self, divisor):
            quotient, remainder = Polynomial([]), self
            ddeg, dlead = divisor.degree(), divisor.leading_coefficient()
            while remainder.degree() >= ddeg:
                shift = remainder.degree() - ddeg
                term = Polynomial([self.field(0)] * shift + [remainder.leading_coefficient() / dlead])
                quotient += term
                remainder -= term * divisor
            return quotient, remainder

        def __truediv__(self, divisor):
            if divisor.is_zero():
                raise ZeroDivisionError
            return divmod(self, divisor)[0]

        def __mod__(self, divisor):
            if divisor.is_zero():
                raise ZeroDivisionError
            return divmod(self, divisor)[1]

        # -- interpolation ---------------------------------------------------
        @classmethod
        def interpolate_at(cls, shares, x_recomb=field(0)):
            """Lagrange value at x_recomb from (x, y) pairs (reference :68-80)."""
            if type(x_recomb) is int:
                x_recomb = field(x_recomb)
            assert type(x_recomb) is GFElement
            xs, ys = zip(*shares)
            weights = []
            for i, x_i in enumerate(xs):
                terms = [(x_k - x_recomb) / (x_k - x_i) for k, x_k in enumerate(xs) if k != i]
                weights.append(reduce(operator.mul, terms))
            return sum(map(operator.mul, ys, weights))

        _lagrange_cache = {}

        @classmethod
        def interpolate(cls, shares):
            """Lagrange interpolation with cached basis polynomials (reference :85-108)."""
            ident, one = cls([field(0), field(1)]), cls([field(1)])
            xs, ys = zip(*shares)

            def basis(xi):
                key = (xs, xi)
                if key not in cls._lagrange_cache:
                    num = reduce(operator.mul, [ident - cls([xj]) for xj in xs if xj != xi], one)
                    den = reduce(operator.mul, [xi - xj for xj in xs if xj != xi], field(1))
                    cls._lagrange_cache[key] = num * cls([1 / den])
                return cls._lagrange_cache[key]

            total = cls([0])
            for xi, yi in zip(xs, ys):
                total += cls([yi]) * basis(xi)
            return total

        @classmethod
        def interpolate_fft(cls, ys, omega):
            """f with f(omega^i) = ys[i], len(ys) a power of two (reference :111-122)."""
            n = len(ys)
            assert n & (n - 1) == 0, "n must be power of two"
            assert type(omega) is GFElement
            assert omega ** n == 1, "must be an n'th root of unity"
            assert omega ** (n // 2) != 1, "must be a primitive n'th root of unity"
            return cls([b / n for b in fft_helper(ys, 1 / omega, field)])

        def evaluate_fft(self, omega, n):
            assert n & (n - 1) == 0, "n must be power of two"
            assert type(omega) is GFElement
            assert omega ** n == 1, "must be an n'th root of unity"
            assert omega ** (n // 2) != 1, "must be a primitive n'th root of unity"
            return fft(self, omega, n)

        @classmethod
        def random(cls, degree, y0=None):
            coeffs = [field.random() for _ in range(degree + 1)]
            if y0 is not None:
                if type(y0) is int:
                    y0 = field(y0)
                assert type(y0) is GFElement
                coeffs[0] = y0
            return cls(coeffs)

        @classmethod
        def interp_extrap(cls, xs, omega):
            """Interpolate on the even powers omega^(2i), evaluate on all omega^i (reference :141-158)."""
            n = len(xs)
            assert n & (n - 1) == 0, "n must be power of 2"
            assert pow(omega, 2 * n) == 1, "omega must be 2n'th root of unity"
            assert pow(omega, n) != 1, "omega must be primitive 2n'th root of unity"
            return cls.interpolate_fft(xs, omega ** 2).evaluate_fft(omega, 2 * n)

        @classmethod
        def interp_extrap_cpp(cls, xs, omega):
            """Same through the native boundary -- here the HIP kernels (reference :161-178)."""
            n = len(xs)
            assert n & (n - 1) == 0, "n must be power of 2"
            assert pow(omega, 2 * n) == 1, "omega must be 2n'th root of unity"
            assert pow(omega, n) != 1, "omega must be primitive 2n'th root of unity"
            p = omega.modulus
            poly = fft_interpolate_cpp(list(range(n)), xs, (omega ** 2).value, p, n)
            return fft_cpp(poly, omega.value, p, 2 * n)

    _poly_cache[field] = Polynomial
    return Polynomial


def _is_pow2(n):
    return n >= 1 and (n & (n - 1)) == 0


def get_omega(field, n, seed=None):
    """A primitive n-th root of unity of the field, n a power of two.

    Behaviour pinned by the reference (polynomial.py:253-268) and by tests/golden/constants.json:
    the candidate is field.random(seed) ** ((p - 1) / n); with seed=0 every party derives the same
    root.  A candidate whose order is a proper divisor of n is thrown away and the draw is repeated
    WITHOUT a seed, exactly as the reference's retry does."""
    if not _is_pow2(n):
        raise AssertionError("n must be a power of 2")
    cofactor, rem = divmod(field.modulus - 1, n)
    if rem:
        raise AssertionError("the field has no element of order n")
    draw_seed = seed
    while True:
        cand = field.random(draw_seed) ** cofactor
        # cand ** n == 1 always; cand is primitive iff its (n/2)-th power is -1 (or n == 1)
        if n == 1 or (cand != 1 and cand ** (n // 2) != 1):
            return cand
        draw_seed = None


def fft_helper(a, omega, field):
    """[sum_j a[j] omega^(i j) for i < len(a)], len(a) a power of two: iterative decimation in time
    over a bit-reversed copy with one table of the n/2 twiddles (same values as the reference's recursive
    helper, polynomial.py:271-292; this is the host-side check of the device NTT, hb_ntt.hip)."""
    n = len(a)
    if not _is_pow2(n):
        raise AssertionError("n must be a power of 2")
    if n == 1:
        return list(a)
    bits = n.bit_length() - 1
    vals = [None] * n
    for j, coeff in enumerate(a):
        vals[int(format(j, "0%db" % bits)[::-1], 2)] = coeff if isinstance(coeff, GFElement) else field(coeff)
    tw = [field(1)]
    for _ in range(n // 2 - 1):
        tw.append(tw[-1] * omega)
    span = 1
    while span < n:
        step = n // (2 * span)
        for start in range(0, n, 2 * span):
            for j in range(span):
                lo, hi = start + j, start + j + span
                t = vals[hi] * tw[j * step]
                vals[hi] = vals[lo] - t
                vals[lo] = vals[lo] + t
        span *= 2
    return vals


def fft(poly, omega, n):
    """Evaluations of poly at omega^0 .. omega^(n-1), omega a primitive n-th root (reference :295-302)."""
    if not _is_pow2(n):
        raise AssertionError("n must be a power of 2")
    field = poly.field
    if len(poly.coeffs) > n or omega ** n != 1 or (n > 1 and omega ** (n // 2) == 1):
        raise AssertionError("need deg < n and omega a primitive n-th root of unity")
    return fft_helper(list(poly.coeffs) + [field(0)] * (n - len(poly.coeffs)), omega, field)


def fnt_decode_step1(poly, zs, omega2, n):
    """Point-set half of the Soro-Lacan FNT interpolation (reference :305-344; C++ twin
    rsdecode_impl.h:194-222).  With x_i = omega^zs[i], omega = omega2^2 and A(X) = prod (X - x_i):
    returns (A evaluated at the 2n powers of omega2, [A'(x_i)]) -- A'(x_i) = prod_{j != i}(x_i - x_j).
    A is built coefficient-wise, transformed once (deg A = k <= n < 2n) and differentiated formally."""
    field = omega2.field
    points = [omega2 ** (2 * z) for z in zs]
    coeffs = [field(1)]
    for x in points:                                  # multiply by (X - x)
        nxt = [field(0)] * (len(coeffs) + 1)
        for j, c in enumerate(coeffs):
            nxt[j + 1] = nxt[j + 1] + c
            nxt[j] = nxt[j] - x * c
        coeffs = nxt
    a_evals = fft_helper(coeffs + [field(0)] * (2 * n - len(coeffs)), omega2, field)
    deriv = [coeffs[j] * j for j in range(1, len(coeffs))]
    a_prime = []
    for x in points:
        acc = field(0)
        for c in reversed(deriv):                     # Horner
            acc = acc * x + c
        a_prime.append(acc)
    return a_evals, a_prime


def fnt_decode_step2(poly, zs, ys, as_, ais_, omega2, n):
    """Value half (reference :347-382; rsdecode_impl.h:226-265): the P of degree < k with
    P(omega^zs[i]) = ys[i].  P / A = sum_i w_i / (X - x_i) with w_i = y_i / A'(x_i), whose power series at 0 has
    coefficients -N(omega^-(m+1)), N(X) = sum_i w_i X^zs[i]; so P = (that series, truncated) * A mod X^k.  One size-n
    transform gives the series, the product is taken through the 2n evaluation points of step 1."""
    field = omega2.field
    k = len(ys)
    if k != len(ais_) or len(as_) != 2 * n:
        raise AssertionError("step-1 tables do not match these points")
    omega = omega2 * omega2
    weights = [field(0)] * n
    for z, y, d in zip(zs, ys, ais_):
        weights[z] = y / d
    n_at = fft_helper(weights, omega, field)                  # N(omega^j), j < n
    series = [-n_at[(n - 1 - m) % n] for m in range(n)]       # -N(omega^-(m+1))
    s_evals = fft_helper(series + [field(0)] * n, omega2, field)
    prod = fft_helper([s * a for s, a in zip(s_evals, as_)], 1 / omega2, field)
    scale = 1 / field(2 * n)
    return poly([c * scale for c in prod[:k]])


class EvalPoint(object):
    """Evaluation-point policy shared by all parties (reference :385-423):
    party i sits at i+1, or at omega^i where omega has order = next power of two >= n."""

    def __init__(self, field, n, use_omega_powers=False):
        self.use_omega_powers = use_omega_powers
        self.field = field
        self.n = n
        if use_omega_powers:
            self.order = n if n & (n - 1) == 0 else 2 ** n.bit_length()
            # every party must derive the same omega: seed 0 (reference :406-410)
            self.omega2 = get_omega(field, 2 * self.order, seed=0)
            self.omega = self.omega2 ** 2
        else:
            self.order = n
            self.omega2 = None
            self.omega = None

    def __call__(self, i):
        if self.use_omega_powers:
            return self.field(pow(self.omega2.value, 2 * i, self.field.modulus))
        return self.field(i + 1)

    def zero(self):
        return self.field(0)
