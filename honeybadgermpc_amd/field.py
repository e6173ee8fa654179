"""
GF(p) as Python objects: the element type of the public API
(reference: honeybadgermpc/field.py:41-289).

Elements are plain Python ints reduced mod p.  They only appear at the edges of the hot
path (batch_reconstruction strips them to ints on entry and re-wraps on exit,
reference batch_reconstruction.py:126,227); all bulk arithmetic happens on the GPU.
"""
from random import Random


class FieldsNotIdentical(Exception):
    pass


class FieldElement(object):
    """Common base class of field element types (reference field.py:32-38)."""

    def __int__(self):
        return self.value

    __long__ = __int__


_SMALL_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def is_probable_prime(n):
    """Miller-Rabin with the first 12 prime bases (deterministic below 3.3e24, and the
    error bound 4^-12 beyond; the reference defers to gmpy2.is_prime, field.py:25,55)."""
    if n < 2:
        return False
    for q in _SMALL_PRIMES:
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in _SMALL_PRIMES:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


class GF(object):
    """Prime field, one object per modulus (multiton, reference field.py:41-65)."""

    _field_cache = {}

    def __new__(cls, modulus):
        return GF._field_cache.setdefault(modulus, super(GF, cls).__new__(cls))

    def __init__(self, modulus):
        if not is_probable_prime(int(modulus)):
            raise ValueError(f"{modulus} is not a prime")
        self.modulus = modulus

    def __call__(self, value):
        return GFElement(value, self)

    def __reduce__(self):
        return (GF, (self.modulus,))

    def random(self, seed=None):
        # Mersenne-Twister draw: get_omega's determinism depends on this exact call
        # (reference field.py:64-65, polynomial.py:262-263)
        return GFElement(Random(seed).randint(0, self.modulus - 1), self)


def _coerce(self, other):
    """int operand of `other`, raising when it belongs to a different field."""
    if isinstance(other, GFElement):
        if other.field is not self.field:
            raise FieldsNotIdentical
        return other.value
    return other


class GFElement(FieldElement):
    __slots__ = ("modulus", "field", "value")

    def __init__(self, value, gf):
        self.modulus = gf.modulus
        self.field = gf
        self.value = value % self.modulus

    # -- ring operations ----------------------------------------------------
    def __add__(self, other):
        if not isinstance(other, (GFElement, int)):
            return NotImplemented
        return GFElement(self.value + _coerce(self, other), self.field)

    __radd__ = __add__

    def __sub__(self, other):
        if not isinstance(other, (GFElement, int)):
            return NotImplemented
        return GFElement(self.value - _coerce(self, other), self.field)

    def __rsub__(self, other):
        return GFElement(other - self.value, self.field)

    def __mul__(self, other):
        if not isinstance(other, (GFElement, int)):
            return NotImplemented
        return GFElement(self.value * _coerce(self, other), self.field)

    __rmul__ = __mul__

    def __pow__(self, exponent):
        return GFElement(pow(self.value, exponent, self.modulus), self.field)

    def __neg__(self):
        return GFElement(-self.value, self.field)

    def __invert__(self):
        if self.value == 0:
            raise ZeroDivisionError("Cannot invert zero")
        return GFElement(pow(self.value, -1, self.modulus), self.field)

    def __div__(self, other):
        if isinstance(other, GFElement):
            if other.field is not self.field:
                raise FieldsNotIdentical
            return self * ~other
        return self * ~GFElement(other, self.field)

    __truediv__ = __div__
    __floordiv__ = __div__

    def __rdiv__(self, other):
        return GFElement(other, self.field) / self

    __rtruediv__ = __rdiv__
    __rfloordiv__ = __rdiv__

    def sqrt(self):
        """A square root (either one).  Reference field.py:170-208."""
        p = self.modulus
        assert p % 2 == 1, "Modulus must be odd"
        assert pow(self.value, (p - 1) // 2, p) == 1
        if p % 4 == 3:
            return GFElement(pow(self.value, (p + 1) // 4, p), self.field)
        # Cipolla: find t with t^2 - a a non-residue, then (t + sqrt(t^2-a))^((p+1)/2)
        a = self.value
        t = 1
        while pow((t * t - a) % p, (p - 1) // 2, p) != p - 1:
            t += 1
        w = (t * t - a) % p

        def mul(u, v):
            return ((u[0] * v[0] + u[1] * v[1] * w) % p, (u[0] * v[1] + u[1] * v[0]) % p)

        result, base, e = (1, 0), (t, 1), (p + 1) // 2
        while e:
            if e & 1:
                result = mul(result, base)
            base = mul(base, base)
            e >>= 1
        return GFElement(result[0], self.field)

    # -- representation / comparison -----------------------------------------
    def bit(self, index):
        return (self.value >> index) & 1

    def signed(self):
        return self.value - self.modulus if self.value > (self.modulus - 1) // 2 else self.value

    def unsigned(self):
        return self.value

    def __repr__(self):
        return "{%d}" % self.value

    __str__ = __repr__

    def __eq__(self, other):
        if isinstance(other, GFElement):
            if other.field is not self.field:
                raise FieldsNotIdentical
            return self.value == other.value
        return self.value == other

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self.field, self.value))

    def __bool__(self):
        return self.value != 0


def fake_gf(modulus):
    """Benchmark-only fake field whose every operation returns p-1 (reference field.py:292-363)."""
    sentinel = modulus - 1

    class FakeFieldElement(FieldElement):
        def __init__(self, value):
            self.value = value

        def _const(self, *_):
            return FakeFieldElement(sentinel)

        __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = _const
        __div__ = __rdiv__ = __truediv__ = __rtruediv__ = __floordiv__ = __rfloordiv__ = _const
        __pow__ = __neg__ = __invert__ = sqrt = _const

        def bit(self, index):
            return 1

        def __repr__(self):
            return "{{%d}}" % self.value

        __str__ = __repr__

    FakeFieldElement.field = FakeFieldElement
    FakeFieldElement.modulus = modulus
    return FakeFieldElement
