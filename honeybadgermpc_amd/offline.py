"""
Offline-phase encodes on device tensors: the same kernels as the batch open, large
batches, no new arithmetic.

* ShareDealer       -- the trusted dealer's share generation: k polynomials of degree t evaluated at the n
                       party points, party-major so that row i is party i's file
                       (reference preprocessing.py:211-239 `_write_polys`; offline_randousha.py:41-53).
* HyperInvertible   -- RanDouSha's refinement: the n parties' contributions are the coefficients of a
                       degree n-1 polynomial, evaluated at the n party points
                       (reference offline_randousha.py:73-78), and the checkers' degree / secret test
                       (reference :95-123).
* random_elements   -- uniform field elements drawn on the device (rejection from 2^bits).

* extract_at_omega_powers -- the randomness extractor of progs/random_refinement.py for any number of contribution vectors.

Element layout as in honeybadgermpc_amd.device: int64 tensors (count, 4), little-endian limbs.
"""
import ctypes

from ._capi import Context, HbView, np_ptr
from .device import BatchOpen


def random_elements(modulus, count, generator=None, device=None):
    """count uniform residues mod `modulus` as a (count, 4) limb tensor: bits-wide draws, rejected when >= modulus."""
    ctx = Context.get(modulus, device)
    t = ctx.torch
    bits = modulus.bit_length()
    top = (bits - 1) // 64                    # highest limb in use
    top_mask = (1 << (bits - 64 * top)) - 1
    nl = ctx.n_limbs
    p_limbs = [(modulus >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(nl)]
    sign = 1 << 63

    def as_i64(v):                            # the int64 with the same bits
        return v - (1 << 64) if v >= sign else v

    out, have = [], 0
    while have < count:
        want = max(1024, int((count - have) * 1.3) + 16)
        cand = t.randint(-(1 << 63), (1 << 63) - 1, (want, nl), dtype=t.int64, device=ctx.tdev, generator=generator)
        cand[:, top] &= as_i64(top_mask)
        for j in range(top + 1, nl):
            cand[:, j] = 0
        # unsigned lexicographic cand < p from the top limb down (x ^ sign turns unsigned order into signed order)
        less = t.zeros(want, dtype=t.bool, device=ctx.tdev)
        equal = t.ones(want, dtype=t.bool, device=ctx.tdev)
        for j in range(nl - 1, -1, -1):
            cj = cand[:, j] ^ as_i64(sign)
            pj = as_i64(p_limbs[j] ^ sign)
            less |= equal & (cj < pj)
            equal &= cj == pj
        keep = cand[less]
        out.append(keep[: count - have])
        have += out[-1].shape[0]
    return t.cat(out, dim=0).contiguous()


class ShareDealer:
    """deal(coeffs): [k][t+1] coefficient rows (flat (k (t+1), 4) tensor) -> [n][k] shares, row i = party i.

    It is the R1 encode of the batch open: the matrix-core path when the shape qualifies."""

    def __init__(self, modulus, n, t, max_polys=1 << 16, device=None):
        self.n, self.t, self.d = n, t, t + 1
        self.op = BatchOpen(modulus, n, t, max_shares=max_polys * (t + 1), device=device)
        self.ctx = self.op.ctx

    def deal(self, coeffs, out=None):
        assert coeffs.shape[0] % self.d == 0
        return self.op.r1_encode(coeffs, out=out)

    def deal_secrets(self, secrets, generator=None):
        """Random degree-t polynomials with the given constant terms (a (k, 4) tensor) -> ([n][k] shares, coeffs)."""
        k = secrets.shape[0]
        coeffs = random_elements(self.ctx.modulus, k * self.d, generator, self.ctx.device).view(k, self.d, self.ctx.n_limbs)
        coeffs[:, 0, :] = secrets
        coeffs = coeffs.reshape(k * self.d, self.ctx.n_limbs).contiguous()
        return self.deal(coeffs), coeffs


class HyperInvertible:
    """RanDouSha's two codec steps for one party, k values per party.

    refine(received):  received[s][j] = the share sender s dealt us for its j-th value  ([n][k], party-major)
                       -> [n][k]: row i = the refined sharing i of every column, i.e. the polynomial with
                       coefficients received[0..n-1][j] evaluated at point(i).
    check(shares, degree): shares[s][j] = party s's share of checked value j ([n][k]) -> (ok, secrets):
                       every interpolated polynomial has exactly the given degree; secrets = constant terms.
    """

    def __init__(self, modulus, n, device=None):
        from .field import GF
        from .polynomial import EvalPoint

        self.ctx = ctx = Context.get(modulus, device)
        self.n = n
        point = EvalPoint(GF(modulus), n, use_omega_powers=False)
        self.x = [point(i).value for i in range(n)]
        self._xh = ctx.host_elems(self.x)
        m = ctypes.c_void_p()
        ctx.check(ctx.lib.hb_vand_matrix_create(ctx.h, np_ptr(self._xh), n, n, ctypes.byref(m), ctx.stream()), "hb_vand_matrix_create")
        self._v = m

    def refine(self, received, out=None):
        received = self.ctx.elems(received, what="received")
        n, k = self.n, received.shape[0] // self.n
        if received.shape[0] != n * k:
            raise ValueError("received: expected n rows of k elements")
        if out is None:
            out = self.ctx.empty(n * k)
        view = HbView(1, k)                   # element (column j, row l) at l * k + j, for the input and the output
        rc = self.ctx.lib.hb_matvec(self.ctx.h, self._v, self.ctx.ptr(received), view, None, self.ctx.ptr(out), view, k, self.ctx.stream())
        self.ctx.check(rc, "hb_matvec")
        return out

    def check(self, shares, degree):
        shares = self.ctx.elems(shares, what="shares")
        t = self.ctx.torch
        n, k = self.n, shares.shape[0] // self.n
        rows = shares.view(n, k, self.ctx.n_limbs).transpose(0, 1).contiguous().view(k * n, self.ctx.n_limbs)     # [k][n]: one polynomial's points per row
        coeffs = self.ctx.empty(k * n)
        rc = self.ctx.lib.hb_vandermonde_batch_interpolate(self.ctx.h, np_ptr(self._xh), n, self.ctx.ptr(rows), k, self.ctx.ptr(coeffs), self.ctx.stream())
        self.ctx.check(rc, "hb_vandermonde_batch_interpolate")
        nz = (coeffs.view(k, n, self.ctx.n_limbs) != 0).any(dim=2)                                 # [k][n] coefficient is non-zero
        ok = bool(nz[:, degree].all().item()) and not bool(nz[:, degree + 1 :].any().item())
        return ok, coeffs.view(k, n, self.ctx.n_limbs)[:, 0, :].contiguous()

    def __del__(self):
        try:
            self.ctx.lib.hb_matrix_destroy(self._v)
        except Exception:
            pass


def extract_at_omega_powers(field, n, t, batches):
    """Randomness extraction (reference progs/random_refinement.py:5-19) for a list of contribution vectors.

    Each vector holds k contributions (n - t <= k <= n, n >= 3 t + 1), read as the coefficients of a polynomial; the result
    keeps its values at the first k - t omega points.  All vectors of one length go through ONE batched encode."""
    from .polynomial import EvalPoint
    from .reed_solomon import EncoderFactory

    if n < 3 * t + 1:
        raise AssertionError(f"randomness extraction needs n >= 3 t + 1 (n = {n}, t = {t})")
    for vec in batches:
        if not n - t <= len(vec) <= n:
            raise AssertionError(f"randomness extraction: {len(vec)} contributions with n = {n}, t = {t}")
    codec = EncoderFactory.get(EvalPoint(field, n, use_omega_powers=True))
    out = [None] * len(batches)
    for k in sorted({len(vec) for vec in batches}):
        which = [i for i, vec in enumerate(batches) if len(vec) == k]
        for i, row in zip(which, codec.encode([list(batches[i]) for i in which])):
            out[i] = row[: k - t]
    return out
