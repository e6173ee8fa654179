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


def get_omega(field, n, seed=None):
    """An n-th root of unity, n a power of two; deterministic for a given seed
    (reference :253-268, including its seed-less retry)."""
    assert n & n - 1 == 0, "n must be a power of 2"
    x = field.random(seed)
    y = pow(x, (field.modulus - 1) // n)
    if y == 1 or pow(y, n // 2) == 1:
        return get_omega(field, n)
    assert pow(y, n) == 1, "omega must be 2n'th root of unity"
    assert pow(y, n // 2) != 1, "omega must be primitive 2n'th root of unity"
    return y


def fft_helper(a, omega, field):
    """Recursive radix-2 transform of the coefficient list a at omega^0..omega^(n-1)
    (reference :271-292).  Host Python: the independent check for the kernels."""
    n = len(a)
    assert not (n & (n - 1)), "n must be a power of 2"
    if n == 1:
        return a
    even = fft_helper(a[0::2], pow(omega, 2), field)
    odd = fft_helper(a[1::2], pow(omega, 2), field)
    half = n // 2
    out = [field(1)] * n
    for j in range(n):
        out[j] = even[j % half] + pow(omega, j) * odd[j % half]
    return out


def fft(poly, omega, n):
    assert n & n - 1 == 0, "n must be a power of 2"
    assert len(poly.coeffs) <= n
    assert pow(omega, n) == 1
    assert pow(omega, n // 2) != 1
    padded = poly.coeffs + [poly.field(0)] * (n - len(poly.coeffs))
    return fft_helper(padded, omega, poly.field)


def fnt_decode_step1(poly, zs, omega2, n):
    """Per-point-set precomputation of Soro-Lacan FNT decoding, Python version
    (reference :305-344): A(X) at the 2n powers of omega2 and A_i(x_i) = prod_{j!=i}(x_i-x_j)."""
    k = len(zs)
    omega = omega2 ** 2
    xs = [omega ** z for z in zs]
    a_ = poly([1])
    for x in xs:
        a_ *= poly([-x, 1])
    as_ = [a_(omega2 ** i) for i in range(2 * n)]
    ais_ = []
    for i in range(k):
        prod = a_.field(1)
        for j in range(k):
            if i != j:
                prod *= xs[i] - xs[j]
        ais_.append(prod)
    return as_, ais_


def fnt_decode_step2(poly, zs, ys, as_, ais_, omega2, n):
    """P with P(omega^zs[i]) = ys[i], O(n log n) given step 1 (reference :347-382)."""
    k = len(ys)
    assert len(ys) == len(ais_)
    assert len(as_) == 2 * n
    omega = omega2 ** 2
    ncoeffs = [0] * n
    for i in range(k):
        ncoeffs[zs[i]] = ys[i] / ais_[i]
    nevals = poly(ncoeffs).evaluate_fft(omega, n)
    power_a = -poly(nevals[::-1])
    pas = power_a.evaluate_fft(omega2, 2 * n)
    prec = poly.interpolate_fft([p * a for p, a in zip(pas, as_)], omega2)
    prec.coeffs = prec.coeffs[:k]
    return prec


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
