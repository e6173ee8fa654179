"""The plan-free robust path (hb_quick.hip) through the C ABI: hb_quick_interp_check against the oracle's interpolate / evaluate
and hb_probe_* against the oracle's Gao (reference rsdecode_impl.h:281-363, reed_solomon.py:151-186) -- bit-exact outcomes."""
import ctypes
import random

import numpy as np
import pytest

import oracle
from conftest import BLS as P, set_hook

pytestmark = pytest.mark.gpu

INT_MAX = (1 << 31) - 1


def _quick(ctx, x, z, zc, cols, c, store=True, lo=0, hi=None):
    import torch

    from honeybadgermpc_amd._capi import np_ptr

    d = len(z)
    out = ctx.empty(c * d) if store else None
    status = torch.tensor([0, INT_MAX], dtype=torch.int32, device="cuda")
    za, zca = np.array(z, dtype=np.int32), np.array(zc if zc else [0], dtype=np.int32)
    rc = ctx.lib.hb_quick_interp_check(ctx.h, np_ptr(ctx.host_elems(x)), len(x), np_ptr(za), d, np_ptr(zca), len(zc), ctx.ptr(cols), c, lo, c if hi is None else hi,
                                       ctx.ptr(out) if store else None, ctx.ptr(status), ctx.stream())
    ctx.check(rc, "hb_quick_interp_check")
    st = status.cpu().tolist()
    return out, st[0], st[1]


@pytest.mark.parametrize("n,t,c,omega", [(16, 5, 300, False), (64, 21, 2000, False), (64, 21, 777, True), (100, 33, 500, False), (256, 85, 260, True), (7, 3, 40, False),
                                        (384, 127, 40, False), (200, 64, 60, False), (130, 42, 50, False), (381, 126, 30, False)])
def test_quick_interp_check_vs_oracle(n, t, c, omega):
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint

    ctx = Context.get(P)
    rnd = random.Random(n * 1000 + t)
    d = t + 1
    point = EvalPoint(GF(P), n, use_omega_powers=omega)
    x = [point(i).value for i in range(n)]
    from structured import structured_rows

    polys = structured_rows(rnd, P, c, d)                                      # zero, constant, short, padded polynomials among the uniform ones
    enc = oracle.vandermonde_batch_evaluate(x, polys, P)                       # [c][n]
    flat = [enc[k][j] for j in range(n) for k in range(c)]
    cols = ctx.upload_ints(flat)
    for trial in range(4):
        order = list(range(n))
        rnd.shuffle(order)
        z = order[:d]
        zc = order[d : d + rnd.choice([0, 1, t, min(n - d, t + 3)])]
        out, flag, first = _quick(ctx, x, z, zc, cols, c)
        got = ctx.download_ints(out)
        assert got == [v for row in polys for v in row], (trial, z, zc)
        assert flag == 0 and first == INT_MAX
        if zc:
            # validate only (no coefficient store) and a corrupted compared column: the flag and the FIRST disagreeing chunk
            _, flag, _ = _quick(ctx, x, z, zc, cols, c, store=False)
            assert flag == 0
            bad_chunks = sorted(rnd.sample(range(c), 3))
            bad = list(flat)
            for m in bad_chunks:
                j = rnd.choice(zc)
                bad[j * c + m] = (bad[j * c + m] + 1 + rnd.randrange(P - 1)) % P
            out, flag, first = _quick(ctx, x, z, zc, ctx.upload_ints(bad), c)
            assert flag == 1 and first == bad_chunks[0], (first, bad_chunks)
            assert ctx.download_ints(out) == [v for row in polys for v in row]       # the decode itself only reads the rows z
        if zc and c > 40:
            # from chunk lo on: earlier chunks are neither read nor written, the first disagreement is counted from lo
            lo = c // 3
            bad = list(flat)
            bad[zc[0] * c + lo - 1] = (bad[zc[0] * c + lo - 1] + 1) % P           # before lo: must not be seen
            bad[zc[-1] * c + lo + 7] = (bad[zc[-1] * c + lo + 7] + 1) % P
            out, flag, first = _quick(ctx, x, z, zc, ctx.upload_ints(bad), c, lo=lo)
            assert flag == 1 and first == 7
            assert ctx.download_ints(out[lo * d :]) == [v for row in polys[lo:] for v in row]
            # a single chunk: [lo + 7, lo + 8) holds the corrupted symbol, [lo + 8, lo + 9) does not; nothing else is written
            out1, flag, first = _quick(ctx, x, z, zc, ctx.upload_ints(bad), c, lo=lo + 7, hi=lo + 8)
            assert flag == 1 and first == 0 and ctx.download_ints(out1[(lo + 7) * d : (lo + 8) * d]) == polys[lo + 7]
            _, flag, _ = _quick(ctx, x, z, zc, ctx.upload_ints(bad), c, lo=lo + 8, hi=lo + 9)
            assert flag == 0
        # a corrupted DECODED column changes coefficients and must disagree with every compared row of those chunks
        if zc:
            bad = list(flat)
            m = rnd.randrange(c)
            bad[z[0] * c + m] = (bad[z[0] * c + m] + 5) % P
            _, flag, first = _quick(ctx, x, z, zc, ctx.upload_ints(bad), c)
            assert flag == 1 and first == m


def test_quick_rejects_what_it_cannot_take():
    from honeybadgermpc_amd._capi import HB_OK, Context, np_ptr

    ctx = Context.get(P)
    x = list(range(1, 9))
    cols = ctx.upload_ints([1] * 8 * 4)
    z = np.array([0, 1, 1, 2], dtype=np.int32)                                   # repeated index
    zc = np.array([5], dtype=np.int32)
    import torch

    st = torch.tensor([0, INT_MAX], dtype=torch.int32, device="cuda")
    assert ctx.lib.hb_quick_interp_check(ctx.h, np_ptr(ctx.host_elems(x)), 8, np_ptr(z), 4, np_ptr(zc), 1, ctx.ptr(cols), 4, 0, 4, None, ctx.ptr(st), ctx.stream()) != HB_OK
    z = np.array([0, 1, 2, 5], dtype=np.int32)                                   # zc overlaps z
    assert ctx.lib.hb_quick_interp_check(ctx.h, np_ptr(ctx.host_elems(x)), 8, np_ptr(z), 4, np_ptr(zc), 1, ctx.ptr(cols), 4, 0, 4, None, ctx.ptr(st), ctx.stream()) != HB_OK
    narrow = Context.get((1 << 61) - 1)
    z = np.array([0, 1, 2, 3], dtype=np.int32)
    colsn = narrow.upload_ints([1] * 32)
    rc = narrow.lib.hb_quick_interp_check(narrow.h, np_ptr(narrow.host_elems(x)), 8, np_ptr(z), 4, np_ptr(zc), 1, narrow.ptr(colsn), 4, 0, 4, None, narrow.ptr(st), narrow.stream())
    assert rc != HB_OK                                                            # UNSUPPORTED: callers fall back to an open plan


class _Probe:
    def __init__(self, ctx, x, k):
        from honeybadgermpc_amd._capi import np_ptr

        self.ctx, self.n = ctx, len(x)
        self.h = ctypes.c_void_p()
        ctx.check(ctx.lib.hb_probe_create(ctx.h, np_ptr(ctx.host_elems(x)), len(x), k, ctypes.byref(self.h), ctx.stream()), "probe_create")

    def feed(self, idx, cols, c, poly, decide=True):
        from honeybadgermpc_amd._capi import np_ptr

        ia = np.array(idx if idx else [0], dtype=np.int32)
        ok = ctypes.c_int32(0)
        mask = np.zeros(self.n, dtype=np.uint8)
        rc = self.ctx.lib.hb_probe_feed(self.h, np_ptr(ia), len(idx), self.ctx.ptr(cols), c, poly, 1 if decide else 0, ctypes.byref(ok), np_ptr(mask), self.ctx.stream())
        self.ctx.check(rc, "probe_feed")
        return (sorted(np.nonzero(mask)[0].tolist()) if ok.value else None) if decide else None

    def reset(self):
        self.ctx.check(self.ctx.lib.hb_probe_reset(self.h), "probe_reset")

    def close(self):
        self.ctx.lib.hb_probe_destroy(self.h)


def _ev(a, x, p):
    r = 0
    for cc in reversed(a):
        r = (r * x + cc) % p
    return r


@pytest.mark.parametrize("p,wgs", [(P, None), ((1 << 61) - 1, None), (257, None), (53, None), (13, None), (P, "4"), (257, "2"), (13, "4"), ((1 << 61) - 1, "3")])
def test_probe_equals_gao_on_every_prefix(monkeypatch, p, wgs):
    """random codewords, random arrival orders, errors inside / at / beyond the radius: after every arrival the probe's verdict
    and its error set equal the oracle's Gao over the same prefix (error set = roots of Gao's locator among all party points).
    wgs: the probe over that many workgroups (HB_PROBE_WGS; what point sets above 128 parties run by default): workgroup 0 the
    coefficients, the others a slice of the value table each, discrepancies and degrees exchanged per point."""
    from honeybadgermpc_amd._capi import Context

    if wgs:
        set_hook(monkeypatch, "HB_PROBE_WGS", wgs)
    ctx = Context.get(p)
    rnd = random.Random(p % 1009)
    trials = decoded = beyond = 0
    for case in range(60 if p > 100 else 150):
        n = rnd.randrange(4, min(40, p))
        k = rnd.randrange(1, n + 1)
        x = rnd.sample(range(1, min(p, 10 ** 6)), n)
        c = rnd.randrange(1, 6)
        poly = rnd.randrange(c)
        arrive = list(range(n))
        rnd.shuffle(arrive)
        arrive = arrive[: rnd.randrange(max(k, 1), n + 1)]
        n1 = len(arrive)
        f = [rnd.randrange(p) for _ in range(k)]
        e = max((n1 - k) // 2, 0)
        nerr = min(rnd.choice([0, 1, e, e, e + 1, e + 1, e + 2, rnd.randrange(0, n1 + 1)]), n1)
        bad = set(rnd.sample(arrive, nerr))
        # the buffer holds c codewords; only `poly` matters, and its values at non-arrived parties are never read meaningfully
        vals = [[rnd.randrange(p) for _ in range(c)] for _ in range(n)]
        for i in range(n):
            vals[i][poly] = rnd.randrange(p) if i in bad else _ev(f, x[i], p)
        cols = ctx.upload_ints([v for row in vals for v in row])
        pr = _Probe(ctx, x, k)
        # first k - 1 points in one batch without a decision, then one at a time
        head = arrive[: max(k - 1, 0)]
        if head:
            pr.feed(head, cols, c, poly, decide=False)
        for m in range(len(head) + 1, n1 + 1):
            got = pr.feed([arrive[m - 1]], cols, c, poly)
            xs = [x[i] for i in arrive[:m]]
            ys = [vals[i][poly] for i in arrive[:m]]
            co, el = oracle.gao_interpolate(xs, ys, k, p)
            want = None
            if co is not None:
                want = sorted(i for i in range(n) if _ev(el, x[i], p) == 0) if len(el) > 1 else []
                beyond += sum(1 for j in range(m) if _ev(co, xs[j], p) != ys[j]) > (m - k) // 2
            assert got == want, (case, n, k, m, got, want)
            trials += 1
            decoded += want is not None
        # reset: the same object decodes another polynomial from scratch, all points at once
        pr.reset()
        got = pr.feed(arrive, cols, c, poly)
        co, el = oracle.gao_interpolate([x[i] for i in arrive], [vals[i][poly] for i in arrive], k, p)
        assert (got is None) == (co is None)
        pr.close()
    assert trials > 300 and decoded > 100 and (beyond >= 1 or p > 100), (trials, decoded, beyond)   # beyond-radius decodes only happen in small fields


def test_probe_at_config3_shape_with_21_liars():
    """n = 64, t = 21: 21 garbage columns first -- every prefix from 43 to 63 points fails, 64 decodes and names the 21 liars"""
    from honeybadgermpc_amd._capi import Context

    ctx = Context.get(P)
    rnd = random.Random(64)
    n, t, c = 64, 21, 3
    d = t + 1
    x = list(range(1, n + 1))
    f = [rnd.randrange(P) for _ in range(d)]
    vals = [[rnd.randrange(P) for _ in range(c)] for _ in range(n)]
    for i in range(n):
        vals[i][1] = rnd.randrange(P) if i < t else _ev(f, x[i], P)
    cols = ctx.upload_ints([v for row in vals for v in row])
    pr = _Probe(ctx, x, d)
    order = list(range(n))
    assert pr.feed(order[:43], cols, c, 1) is None
    for m in range(44, 64):
        assert pr.feed([order[m - 1]], cols, c, 1) is None, m
    assert pr.feed([order[63]], cols, c, 1) == list(range(t))
    pr.close()


@pytest.mark.parametrize("n,t,liars", [(256, 85, "spread"), (200, 66, "first"), (130, 43, "spread")])
def test_probe_over_several_workgroups_at_full_size(n, t, liars):
    """point sets above 128 parties: the probe's launches are three workgroups.  t liars (after every two honest senders, or the first t arrivals),
    d + t points at once and then one at a time: every verdict and error set equal the oracle's Gao over the same prefix; a reset and
    all points in one launch give the last verdict again."""
    from honeybadgermpc_amd._capi import Context

    ctx = Context.get(P)
    rnd = random.Random(n)
    d, c, poly = t + 1, 2, 1
    x = list(range(1, n + 1))
    f = [rnd.randrange(P) for _ in range(d)]
    order = list(range(n))
    rnd.shuffle(order)
    bad = set(order[2::3][:t]) if liars == "spread" else set(order[:t])
    vals = [[rnd.randrange(P) for _ in range(c)] for _ in range(n)]
    for i in range(n):
        vals[i][poly] = rnd.randrange(P) if i in bad else _ev(f, x[i], P)
    cols = ctx.upload_ints([v for row in vals for v in row])
    pr = _Probe(ctx, x, d)

    def want(m):
        co, el = oracle.gao_interpolate([x[i] for i in order[:m]], [vals[i][poly] for i in order[:m]], d, P)
        return None if co is None else (sorted(i for i in range(n) if _ev(el, x[i], P) == 0) if len(el) > 1 else [])

    first = d + t
    assert pr.feed(order[:first], cols, c, poly) == want(first)
    decoded = 0
    for m in range(first + 1, n + 1):
        got = pr.feed([order[m - 1]], cols, c, poly)
        assert got == want(m), m
        decoded += got is not None
    assert decoded >= 1 and got == sorted(bad)
    pr.reset()
    assert pr.feed(order, cols, c, poly) == sorted(bad)
    pr.close()


@pytest.mark.parametrize("p,n,d,c", [(P, 64, 22, 5), (P, 256, 86, 3), (P, 100, 34, 2), ((1 << 64) - 59, 40, 11, 4), (P, 7, 3, 1)])
def test_candidate_check_vs_oracle(p, n, d, c):
    """hb_candidate_check: one polynomial at all n points and, per party, whether its symbol of the chunk differs from that value -- the
    oracle's evaluation, symbols altered at known parties (one word of one limb, the top limb), every chunk of a small buffer in turn"""
    from honeybadgermpc_amd._capi import Context, np_ptr

    ctx = Context.get(p)
    rnd = random.Random(n * 7 + d)
    x = list(range(1, n + 1)) if n != 100 else rnd.sample(range(1, 10 ** 6), n)
    xh = ctx.host_elems(x)
    L = ctx.n_limbs
    for chunk in range(c):
        f = [rnd.randrange(p) for _ in range(d)]
        if chunk == 1:
            f = [0] * d
        want = oracle.vandermonde_batch_evaluate(x, [f], p)[0]
        vals = [[rnd.randrange(p) for _ in range(c)] for _ in range(n)]
        liars = set(rnd.sample(range(n), rnd.randrange(0, n // 2 + 1)))
        for i in range(n):
            vals[i][chunk] = want[i] if i not in liars else (want[i] + 1 + rnd.randrange(p - 1)) % p
        one_bit = rnd.randrange(n)
        if one_bit not in liars:
            liars.add(one_bit)
            vals[one_bit][chunk] = want[one_bit] ^ (1 << (64 * L - 8)) if (want[one_bit] ^ (1 << (64 * L - 8))) < p else (want[one_bit] + 1) % p
        cols = ctx.upload_ints([v for row in vals for v in row])
        coeffs = ctx.upload_ints(f)
        ev = np.empty((n, L), dtype=np.int64)
        diff = np.empty(n, dtype=np.uint8)
        ctx.check(ctx.lib.hb_candidate_check(ctx.h, np_ptr(xh), n, ctx.ptr(coeffs), d, ctx.ptr(cols), c, chunk, np_ptr(ev), np_ptr(diff), ctx.stream()), "hb_candidate_check")
        got = [sum((int(ev[i, j]) & ((1 << 64) - 1)) << (64 * j) for j in range(L)) for i in range(n)]
        assert got == want
        assert set(np.nonzero(diff)[0].tolist()) == liars


def test_point_tables_are_cache_entries():
    """the per-point-set tables (n^2 inverse differences) are entries of the context's bounded table cache: hundreds of point sets
    leave at most the cap resident, a probe keeps its own table alive across a clear, and results stay right"""
    from honeybadgermpc_amd._capi import Context

    ctx = Context.get(P)
    rnd = random.Random(99)
    n, k = 12, 4
    ctx.cache_clear()
    keep = None
    for i in range(260):
        x = rnd.sample(range(1, 10 ** 6), n)
        pr = _Probe(ctx, x, k)
        if i == 0:
            keep = (pr, x)
        else:
            pr.close()
    assert ctx.cache_entries() <= 192 + 8, ctx.cache_entries()
    ctx.cache_clear()
    assert ctx.cache_entries() == 0
    pr, x = keep                                     # its table left the cache; the probe still works on it
    f = [rnd.randrange(P) for _ in range(k)]
    vals = [[_ev(f, x[i], P)] for i in range(n)]
    vals[5][0] = (vals[5][0] + 1) % P
    cols = ctx.upload_ints([v for row in vals for v in row])
    assert pr.feed(list(range(n)), cols, 1, 0) == [5]
    pr.close()


@pytest.mark.parametrize("n,t,c,omega", [(64, 21, 2000, False), (16, 5, 333, True), (100, 33, 1500, False)])
def test_quick_interp_check_map_names_every_disagreeing_chunk(n, t, c, omega):
    """hb_quick_interp_check_map: a bit per chunk (counted from chunk_lo) on which some COMPARED column disagrees -- corrupted interpolation
    columns change the candidates, not covered here -- for whole batches and sub-ranges; flag and first chunk as hb_quick_interp_check"""
    import torch

    from honeybadgermpc_amd._capi import Context, np_ptr
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint

    ctx = Context.get(P)
    rnd = random.Random(n + 31 * c)
    d = t + 1
    point = EvalPoint(GF(P), n, use_omega_powers=omega)
    x = [point(i).value for i in range(n)]
    from structured import structured_rows

    polys = structured_rows(rnd, P, c, d)
    enc = oracle.vandermonde_batch_evaluate(x, polys, P)
    flat = [enc[k][j] for j in range(n) for k in range(c)]
    for trial in range(4):
        order = list(range(n))
        rnd.shuffle(order)
        z, zc = order[:d], order[d : d + t]
        bad_chunks = sorted(rnd.sample(range(c), rnd.choice([1, 2, 17, 40])))
        bad = list(flat)
        for m in bad_chunks:
            for j in rnd.sample(zc, rnd.choice([1, 1, 3])):
                bad[j * c + m] = (bad[j * c + m] + 1 + rnd.randrange(P - 1)) % P
        cols = ctx.upload_ints(bad)
        for lo, hi in [(0, c), (c // 3, c), (c // 5 + 3, c - c // 7), (bad_chunks[0], bad_chunks[0] + 1)]:
            status = torch.tensor([0, INT_MAX], dtype=torch.int32, device="cuda")
            words = torch.zeros((hi - lo + 31) // 32 + 1, dtype=torch.int32, device="cuda")
            out = ctx.empty(c * d)
            za, zca = np.array(z, dtype=np.int32), np.array(zc, dtype=np.int32)
            rc = ctx.lib.hb_quick_interp_check_map(ctx.h, np_ptr(ctx.host_elems(x)), n, np_ptr(za), d, np_ptr(zca), len(zc), ctx.ptr(cols), c, lo, hi,
                                                   ctx.ptr(out), ctx.ptr(status), ctx.ptr(words), ctx.stream())
            ctx.check(rc, "hb_quick_interp_check_map")
            flag, first = status.cpu().tolist()
            bits = np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")
            got = (np.nonzero(bits)[0] + lo).tolist()
            want = [m for m in bad_chunks if lo <= m < hi]
            assert got == want, (trial, lo, hi)
            assert flag == (1 if want else 0) and first == (want[0] - lo if want else INT_MAX)
            assert ctx.download_ints(out[lo * d : hi * d]) == [v for row in polys[lo:hi] for v in row]
