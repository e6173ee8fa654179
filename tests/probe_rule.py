"""Pure-Python statement of the device probe's decision rule (honeybadgermpc_amd/csrc/hb_quick.hip, k_probe_feed): Koetter /
Welch-Berlekamp rational interpolation, one point at a time, fraction-free, and the reference's Gao outcome read off the reduced
basis.  Test infrastructure: pins the RULE against the oracle on the CPU (tests/test_probe_rule.py); the kernel is compared
with the oracle directly in tests/test_gpu_quick.py."""


def deg(a):
    d = len(a) - 1
    while d >= 0 and a[d] == 0:
        d -= 1
    return d


def ev(a, x, p):
    r = 0
    for c in reversed(a):
        r = (r * x + c) % p
    return r


class Probe:
    def __init__(self, k, p, size):
        self.k, self.p = k, p
        self.A = [[1] + [0] * size, [0] * (size + 1)]
        self.B = [[0] * (size + 1), [1] + [0] * size]
        self.pts = []

    def order(self, j):
        """leading monomial of Q_j under the (1, k-1) weighted order, ties: the Y term is the larger"""
        da, db = deg(self.A[j]), deg(self.B[j])
        wa = da if da >= 0 else -1
        wb = db + self.k - 1 if db >= 0 else -1
        return (max(wa, wb), 1 if wb >= wa else 0)

    def add(self, x, y):
        p = self.p
        self.pts.append((x, y))
        dl = [(ev(self.A[j], x, p) + y * ev(self.B[j], x, p)) % p for j in range(2)]
        live = [j for j in range(2) if dl[j]]
        if not live:
            return
        js = min(live, key=self.order)
        for j in live:
            if j != js:
                self.A[j] = [(dl[js] * a - dl[j] * b) % p for a, b in zip(self.A[j], self.A[js])]
                self.B[j] = [(dl[js] * a - dl[j] * b) % p for a, b in zip(self.B[j], self.B[js])]
        for M in (self.A, self.B):
            old = M[js]
            M[js] = [((old[i - 1] if i else 0) - x * old[i]) % p for i in range(len(old))]

    def decide(self, party_points):
        """the reference's Gao outcome for the points so far: None, or the error positions = roots of the locator among
        ALL party points (reed_solomon.py:174-184).  Gao stops at the first remainder of degree < T = (n' + k) // 2
        (rsdecode_impl.h:281-323); in module terms: Q_0 (leading term in A) must have deg A_0 >= T, the row Gao lands on is
        Q_1 reduced against Q_0 to deg A_1 < T, and it decodes iff B_1 divides A_1."""
        p = self.p
        n1 = len(self.pts)
        T = (n1 + self.k) // 2
        j0 = 0 if self.order(0)[1] == 0 else 1
        j1 = 1 - j0
        assert self.order(j0)[1] == 0 and self.order(j1)[1] == 1
        A0, B0, A1, B1 = self.A[j0], self.B[j0], list(self.A[j1]), list(self.B[j1])
        a0 = deg(A0)
        if a0 < T:
            return None
        if a0 == T and deg(A1) >= T:
            assert deg(A1) == T
            c0, c1 = A0[T], A1[T]
            A1 = [(c0 * u - c1 * v) % p for u, v in zip(A1, A0)]
            B1 = [(c0 * u - c1 * v) % p for u, v in zip(B1, B0)]
        assert deg(A1) < T
        db = deg(B1)
        assert db >= 0
        # B1 | A1  <=>  B1 has deg B1 distinct roots among the points fed (A1 + y B1 vanishes on every point; conversely the
        # locator of a decodable word is the product over its errors): no polynomial division
        if sum(1 for x, _ in self.pts if ev(B1, x, p) == 0) != db:
            return None
        return [i for i, x in enumerate(party_points) if ev(B1, x, p) == 0]
