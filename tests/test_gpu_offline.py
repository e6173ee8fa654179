"""
GPU parity for the offline-phase encodes: honeybadgermpc_amd.offline and
progs.triple_refinement against exact Python integers / the oracle.  Bit-exact.
"""
import asyncio
import random

import pytest

import oracle
from conftest import BLS
from structured import structured_rows


def _padded(rows, d):
    """the reference's Welch-Berlekamp decoder strips trailing zeros (reed_solomon_wb.py:121-126 over polynomial.py:14-20): a robust-decoded row of the
    host mirror may be shorter than degree + 1; the device decoder's tensor is zero padded"""
    return [list(r) + [0] * (d - len(r)) for r in rows]

pytestmark = pytest.mark.gpu

P = BLS


def lagrange_at_zero(points, values, p):
    acc = 0
    for i, (xi, yi) in enumerate(zip(points, values)):
        num = den = 1
        for j, xj in enumerate(points):
            if j != i:
                num = num * (-xj) % p
                den = den * (xi - xj) % p
        acc = (acc + yi * num * pow(den, -1, p)) % p
    return acc


def test_random_elements_uniform_range():
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.offline import random_elements

    for p in (P, (1 << 255) - 19, (1 << 64) - 59, 13):
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        vals = Context.get(p).download_ints(random_elements(p, 5000, g))
        assert len(vals) == 5000 and all(0 <= v < p for v in vals)
        if p > 1 << 60:
            assert len(set(vals)) == 5000
            assert max(vals) > p - p // 50 and min(vals) < p // 50     # both ends of the range are reached
        else:
            assert set(vals) == set(range(p))
        g.manual_seed(5)
        assert Context.get(p).download_ints(random_elements(p, 5000, g)) == vals   # reproducible from the seed


@pytest.mark.parametrize("n, t, k", [(4, 1, 9), (7, 2, 50), (16, 5, 333), (64, 21, 100)])
def test_share_dealer_vs_python_ints(n, t, k):
    import torch

    from honeybadgermpc_amd.offline import ShareDealer

    rnd = random.Random(n * 100 + t)
    dealer = ShareDealer(P, n, t, max_polys=k)
    ctx = dealer.ctx
    secrets = [rnd.randrange(P) for _ in range(k)]
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    shares, coeffs = dealer.deal_secrets(ctx.upload_ints(secrets), g)
    cf = ctx.download_ints(coeffs)
    sh = ctx.download_ints(shares)
    assert len(sh) == n * k and len(cf) == k * (t + 1)
    assert [cf[j * (t + 1)] for j in range(k)] == secrets
    # party-major: row i holds party i's share of every value; the oracle evaluates the same polynomials
    want = oracle.vandermonde_batch_evaluate(list(range(1, n + 1)), [cf[j * (t + 1) : (j + 1) * (t + 1)] for j in range(k)], P)
    for i in range(n):
        assert sh[i * k : (i + 1) * k] == [want[j][i] for j in range(k)]
    # any t + 1 parties reconstruct
    parties = rnd.sample(range(n), t + 1)
    for j in rnd.sample(range(k), min(k, 5)):
        assert lagrange_at_zero([i + 1 for i in parties], [sh[i * k + j] for i in parties], P) == secrets[j]


@pytest.mark.parametrize("n, t, k", [(4, 1, 6), (7, 2, 40), (16, 5, 64)])
def test_hyperinvertible_refine_and_check(n, t, k):
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.offline import HyperInvertible

    rnd = random.Random(n + 1000 * t)
    ctx = Context.get(P)
    hi = HyperInvertible(P, n)
    received = [[rnd.randrange(P) for _ in range(k)] for _ in range(n)]          # [sender][value index]
    out = ctx.download_ints(hi.refine(ctx.upload_ints([v for row in received for v in row])))
    for i in range(n):
        for j in range(k):
            assert out[i * k + j] == sum(received[s][j] * pow(i + 1, s, P) for s in range(n)) % P
    # the checkers' test: sharings of exact degree `deg` pass and yield their secrets; any other degree fails
    for deg in (t, 2 * t):
        polys = [[rnd.randrange(P) for _ in range(deg)] + [rnd.randrange(1, P)] for _ in range(k)]
        shares = [[sum(c * pow(i + 1, e, P) for e, c in enumerate(poly)) % P for poly in polys] for i in range(n)]
        flat = ctx.upload_ints([v for row in shares for v in row])
        ok, secrets = hi.check(flat, deg)
        assert ok and ctx.download_ints(secrets) == [poly[0] for poly in polys]
        assert not hi.check(flat, deg + 1)[0]
        if deg > 0:
            assert not hi.check(flat, deg - 1)[0]
        shares[rnd.randrange(n)][rnd.randrange(k)] ^= 1                            # one corrupted share raises the degree
        assert not hi.check(ctx.upload_ints([v for row in shares for v in row]), deg)[0]


class _Bus:
    """In-process stand-in for the MPC runtime's open: collects every party's shares and interpolates at 0."""

    def __init__(self, n, t, field):
        self.n, self.t, self.field = n, t, field
        self.pending = {}

    async def open(self, open_id, party, values):
        slot = self.pending.setdefault(open_id, {"shares": {}, "done": asyncio.get_running_loop().create_future()})
        slot["shares"][party] = values
        if len(slot["shares"]) == self.n:
            p = self.field.modulus
            parties = sorted(slot["shares"])[: self.t + 1]
            pts = [i + 1 for i in parties]
            res = [lagrange_at_zero(pts, [slot["shares"][i][j] for i in parties], p) for j in range(len(values))]
            slot["done"].set_result([self.field(v) for v in res])
        return await slot["done"]


class _ShareArray:
    def __init__(self, ctx, values):
        self.ctx, self.values = ctx, [int(v) % ctx.field.modulus for v in values]

    def __sub__(self, other):
        p = self.ctx.field.modulus
        return _ShareArray(self.ctx, [(a - b) % p for a, b in zip(self.values, other.values)])

    def open(self):
        self.ctx.opens += 1
        return self.ctx.bus.open(self.ctx.opens, self.ctx.myid, self.values)


class _Context:
    def __init__(self, n, t, field, myid, bus):
        self.N, self.t, self.field, self.myid, self.bus, self.opens = n, t, field, myid, bus, 0

    def ShareArray(self, values):  # noqa: N802 (the reference's name)
        return _ShareArray(self, values)


@pytest.mark.parametrize("n, t, k", [(4, 1, 3), (4, 1, 4), (7, 2, 5), (7, 2, 7)])   # reference tests/progs/test_triple_refinement.py:9
def test_triple_refinement(n, t, k):
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.progs.triple_refinement import refine_triples

    rnd = random.Random(n * 10 + k)
    field = GF(P)

    def share(secret):
        poly = [secret] + [rnd.randrange(P) for _ in range(t)]
        return [sum(c * pow(i + 1, e, P) for e, c in enumerate(poly)) % P for i in range(n)]

    triples = [(a, b, a * b % P) for a, b in ((rnd.randrange(P), rnd.randrange(P)) for _ in range(k))]
    sh = [[share(v) for v in tr] for tr in triples]          # [triple][a|b|c][party]

    async def main():
        bus = _Bus(n, t, field)
        ctxs = [_Context(n, t, field, i, bus) for i in range(n)]
        return await asyncio.gather(*[
            refine_triples(ctxs[i], [sh[j][0][i] for j in range(k)], [sh[j][1][i] for j in range(k)], [sh[j][2][i] for j in range(k)])
            for i in range(n)
        ])

    outs = asyncio.run(main())                                 # [party] -> (p, q, pq) share lists
    count = (k - 2 * t + 1) // 2
    assert all(len(o[0]) == len(o[1]) == len(o[2]) == count for o in outs)
    pts = list(range(1, t + 2))
    for idx in range(count):
        vals = [lagrange_at_zero(pts, [outs[i][part][idx] for i in range(t + 1)], P) for part in range(3)]
        assert vals[0] * vals[1] % P == vals[2]
        # the refined values are shared with degree t: every party's share lies on the same polynomial
        for part in range(3):
            for extra in range(t + 1, n):
                sub = list(range(1, t + 1)) + [extra]
                assert lagrange_at_zero([i + 1 for i in sub], [outs[i][part][idx] for i in sub], P) == vals[part]


# ---- device-resident IncrementalDecoder against the host mirror of the reference's ----------
@pytest.mark.parametrize("robust, use_omega", [("gao", False), ("wb", False), ("gao", True)])
@pytest.mark.parametrize("n, t, c, seed", [(4, 1, 1, 1), (4, 1, 7, 2), (7, 2, 5, 3), (7, 2, 40, 4), (10, 3, 33, 5), (16, 5, 64, 6), (16, 5, 9, 7), (13, 4, 300, 8)])
def test_device_incremental_decoder_trajectory(n, t, c, seed, robust, use_omega):
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint
    from honeybadgermpc_amd.reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory

    rnd = random.Random(seed)
    ctx = Context.get(P)
    if robust == "wb" and c > 40:
        c = 40                                          # the host mirror's Welch-Berlekamp is a row reduction per polynomial beyond the radius
    point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
    xs = [point(i).value for i in range(n)]
    codec = Algorithm.FFT if use_omega else Algorithm.VANDERMONDE
    robust_launches = 0
    for trial in range(6):
        polys = structured_rows(rnd, P, c, t + 1)          # zero, constant, short, padded polynomials among the uniform ones
        cols = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
        liars = rnd.sample(range(n), rnd.randrange(trial % 2, t + 1))         # odd trials: at least one liar
        for i in liars:
            kind = rnd.randrange(4)
            hit = {0: range(c), 1: [c - 1], 2: [rnd.randrange(c)], 3: rnd.sample(range(c), max(1, c // 3))}[kind]
            for j in hit:
                cols[i][j] = (cols[i][j] + 1 + rnd.randrange(P - 1)) % P
        order = list(range(n))
        rnd.shuffle(order)
        if trial % 2:                                   # ... who is among the first arrivals, so that it cannot be missed
            order.remove(liars[0])
            order.insert(rnd.randrange(0, t + 1), liars[0])
        host = IncrementalDecoder(EncoderFactory.get(point, codec), DecoderFactory.get(point, codec),
                                  RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO if robust == "gao" else Algorithm.WELCH_BERLEKAMP),
                                  degree=t, batch_size=c, max_errors=t)
        dev = DeviceIncrementalDecoder(P, n, t, batch_size=c, robust=robust, use_omega_powers=use_omega)
        blew_up = False
        for step, idx in enumerate(order):
            try:
                host.add(idx, cols[idx])
            except (AssertionError, Exception) as exc:  # noqa: B014
                # the reference's Welch-Berlekamp robust decoder re-raises what its solver raises beyond what it can decode
                # ("No solution", or the 2t+1+c <= n assertion once confirmed errors shrank the point set,
                # reed_solomon.py:205-212, reed_solomon_wb.py:132,245): the device decoder must fail the same way, there
                assert robust == "wb"
                with pytest.raises(type(exc)):
                    dev.add(idx, ctx.upload_ints(cols[idx]))
                blew_up = True
                break
            dev.add(idx, ctx.upload_ints(cols[idx]) if step % 2 else list(cols[idx]))
            assert dev.done() == host.done(), (trial, step)
            assert dev._confirmed_errors == host._confirmed_errors, (trial, step)
            assert dev._z == host._z and dev._num_decoded == host._num_decoded, (trial, step)
            if host.done():
                hres, herr = host.get_results()
                dres, derr = dev.get_results()
                assert derr == herr
                assert ctx.download_ints(dres.reshape(-1, 4)) == [v for row in _padded(hres, t + 1) for v in row]
                assert _padded(hres, t + 1) == polys               # and both recovered what was shared
                break
        assert blew_up or (host.done() and set(host.get_results()[1]) <= set(liars))
        robust_launches += dev.launches + dev.plan_accepts + dev.probes
    assert robust_launches > 0          # the seeds above all reach the robust path at least once


@pytest.mark.parametrize("robust, liar_count", [("wb", 1), ("wb", 2), ("wb", 5)])
@pytest.mark.parametrize("pattern", ["late-distinct", "everywhere"])
def test_device_decoder_adversaries_welch_berlekamp(robust, liar_count, pattern):
    """the same adversaries with the Welch-Berlekamp robust decoder (n = 16, t = 5): the plan-free path settles what lies inside the
    unique-decoding radius from its batched candidates and hands everything else to hb_wb_decode; the reference's refusals
    ("No solution", the 2t + 1 + c <= n assertion once expelled senders shrank the list) must come at the same column"""
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint
    from honeybadgermpc_amd.reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory

    n, t, c = 16, 5, 30
    rnd = random.Random(liar_count * 17 + len(pattern))
    ctx = Context.get(P)
    point = EvalPoint(GF(P), n)
    xs = [point(i).value for i in range(n)]
    polys = structured_rows(rnd, P, c, t + 1)
    cols = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
    liars = rnd.sample(range(n), liar_count)
    for r, i in enumerate(liars):
        for j in (range(c) if pattern == "everywhere" else [1 + (r * 7) % (c - 1)]):
            cols[i][j] = (cols[i][j] + 1 + rnd.randrange(P - 1)) % P
    honest = [i for i in range(n) if i not in liars]
    rnd.shuffle(honest)
    order = liars + honest
    host = IncrementalDecoder(EncoderFactory.get(point, Algorithm.VANDERMONDE), DecoderFactory.get(point, Algorithm.VANDERMONDE),
                              RobustDecoderFactory.get(t, point, algorithm=Algorithm.WELCH_BERLEKAMP), degree=t, batch_size=c, max_errors=t)
    dev = DeviceIncrementalDecoder(P, n, t, batch_size=c, robust=robust)
    for step, idx in enumerate(order):
        try:
            host.add(idx, cols[idx])
        except (AssertionError, Exception) as exc:  # noqa: B014 - the reference re-raises bare Exceptions
            with pytest.raises(type(exc)):
                dev.add(idx, ctx.upload_ints(cols[idx]))
            return
        dev.add(idx, ctx.upload_ints(cols[idx]))
        assert dev.done() == host.done(), step
        assert dev._confirmed_errors == host._confirmed_errors and dev._z == host._z and dev._num_decoded == host._num_decoded, step
        if host.done():
            break
    assert host.done() and dev.done()
    hres, herr = host.get_results()
    dres, derr = dev.get_results()
    assert derr == herr and ctx.download_ints(dres.reshape(-1, 4)) == [v for row in _padded(hres, t + 1) for v in row] == [v for row in polys for v in row]
    assert dev.quick_launches > 0


@pytest.mark.parametrize("n, t, c, pattern", [(31, 10, 120, "late-distinct"), (31, 10, 120, "late-same"), (64, 21, 96, "late-distinct"), (64, 21, 64, "everywhere"),
                                              (16, 5, 300, "late-distinct")])
def test_device_decoder_worst_case_adversaries(n, t, c, pattern):
    """t liars that arrive FIRST.  "everywhere": garbage in every chunk (every probe up to the last column
    fails).  "late-*": each liar corrupts one chunk k > 0 only -- polynomial 0 decodes clean, so nothing that looks at polynomial
    0 alone can vouch for the batch; "late-distinct" gives every liar a chunk of its own (one round of probe + batched check per
    liar), "late-same" puts them all in one chunk.  The device decoder (plan-free path: hb_quick_interp_check + hb_probe_*) must
    walk the host mirror's trajectory -- done / confirmed errors / arrival list / polynomials decoded -- after EVERY column."""
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint
    from honeybadgermpc_amd.reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory

    rnd = random.Random(n * 100 + t + len(pattern))
    ctx = Context.get(P)
    point = EvalPoint(GF(P), n)
    xs = [point(i).value for i in range(n)]
    polys = structured_rows(rnd, P, c, t + 1)
    cols = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
    liars = rnd.sample(range(n), t)
    for r, i in enumerate(liars):
        if pattern == "everywhere":
            hit = range(c)
        elif pattern == "late-same":
            hit = [c // 2]
        else:
            hit = [1 + (r * 7) % (c - 1)]
        for j in hit:
            cols[i][j] = (cols[i][j] + 1 + rnd.randrange(P - 1)) % P
    honest = [i for i in range(n) if i not in liars]
    rnd.shuffle(honest)
    order = liars + honest
    host = IncrementalDecoder(EncoderFactory.get(point, Algorithm.VANDERMONDE), DecoderFactory.get(point, Algorithm.VANDERMONDE),
                              RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO), degree=t, batch_size=c, max_errors=t)
    dev = DeviceIncrementalDecoder(P, n, t, batch_size=c)
    for step, idx in enumerate(order):
        host.add(idx, cols[idx])
        dev.add(idx, ctx.upload_ints(cols[idx]))
        assert dev.done() == host.done(), step
        assert dev._confirmed_errors == host._confirmed_errors and dev._z == host._z and dev._num_decoded == host._num_decoded, step
        if host.done():
            break
    assert host.done() and dev.done()
    hres, herr = host.get_results()
    dres, derr = dev.get_results()
    assert derr == herr == set(liars)
    assert ctx.download_ints(dres.reshape(-1, 4)) == [v for row in _padded(hres, t + 1) for v in row] == [v for row in polys for v in row]
    assert dev.quick_launches > 0 and dev.launches == 0          # the plan-free path did it: no batched Gao launch, no plan
    # every polynomial that had errors was settled either by the probe or by a batched candidate inside the radius
    if pattern != "late-same":
        # ("everywhere", liars first: the candidate from the newest columns names all t of them at the first polynomial -- one verdict)
        assert dev.probes + dev.radius_verdicts >= (1 if pattern == "everywhere" else len({1 + (r * 7) % (c - 1) for r in range(t)}))


def test_symbols_fetch_matches_the_buffer():
    """hb_symbols_fetch: the symbols of one polynomial in a few columns, through pinned memory -- against plain indexing of the same buffer;
    bad arguments are refused"""
    import numpy as np
    import torch

    from honeybadgermpc_amd._capi import HB_ERR_BAD_ARG, Context, np_ptr

    ctx = Context.get(P)
    rnd = random.Random(17)
    n, c = 70, 37
    cols = ctx.upload_ints([rnd.randrange(P) for _ in range(n * c)]).view(n, c, 4)
    host = cols.cpu().numpy()
    for count in (1, 2, 5, 64):
        for chunk in (0, 11, c - 1):
            idx = np.asarray([rnd.randrange(n) for _ in range(count)], dtype=np.int32)
            out = np.full((count, 4), -1, dtype=np.int64)
            ctx.check(ctx.lib.hb_symbols_fetch(ctx.h, ctx.ptr(cols), n, c, chunk, np_ptr(idx), count, np_ptr(out), ctx.stream()), "hb_symbols_fetch")
            assert (out == host[idx, chunk]).all(), (count, chunk)
    idx = np.zeros(65, dtype=np.int32)
    out = np.zeros((65, 4), dtype=np.int64)
    for count, chunk in ((0, 0), (65, 0), (1, c), (1, -1)):
        assert ctx.lib.hb_symbols_fetch(ctx.h, ctx.ptr(cols), n, c, chunk, np_ptr(idx), count, np_ptr(out), ctx.stream()) == HB_ERR_BAD_ARG
    idx[0] = n
    assert ctx.lib.hb_symbols_fetch(ctx.h, ctx.ptr(cols), n, c, 0, np_ptr(idx), 1, np_ptr(out), ctx.stream()) == HB_ERR_BAD_ARG
    torch.cuda.synchronize()
    # a one-limb context (p < 2^64): one word per symbol
    import ctypes

    from honeybadgermpc_amd._capi import ints_to_limbs, load_library

    lib, p64 = load_library(), (1 << 64) - 59
    h = ctypes.c_void_p()
    assert lib.hb_ctx_create(ctypes.byref(h), np_ptr(ints_to_limbs([p64], p64 + 1, 8)), 1, 0) == 0
    narrow = torch.from_numpy(np.array([rnd.randrange(p64) for _ in range(n * c)], dtype=np.uint64).view(np.int64).copy()).cuda().view(n, c, 1)
    idx = np.asarray([5, 0, n - 1], dtype=np.int32)
    out = np.zeros((3, 1), dtype=np.int64)
    assert lib.hb_symbols_fetch(h, ctypes.c_void_p(narrow.data_ptr()), n, c, 7, np_ptr(idx), 3, np_ptr(out), None) == 0
    assert (out == narrow.cpu().numpy()[idx, 7]).all()
    lib.hb_ctx_destroy(h)


@pytest.mark.parametrize("n, t, c, use_omega, shared", [(16, 5, 7, False, 0), (16, 5, 7, False, 3), (16, 5, 7, False, 5), (16, 5, 4, True, 2), (64, 21, 2, False, 9)])
def test_device_decoder_candidates_against_coordinated_liars(n, t, c, use_omega, shared):
    """The liars all send the values of ONE other polynomial of the right degree (equal to the true one at `shared` honest points, so that
    those honest senders "agree" with the fake too) and arrive first, then t + 1 of them: the word Gao is offered decodes to the fake at some
    prefixes.  The decoder's waiting candidates (device.py _candidate_cap: no incremental decode while a candidate's disagreements stay
    within max_errors - confirmed) must take the reference's decisions column by column -- against the host mirror of IncrementalDecoder."""
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint
    from honeybadgermpc_amd.reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory

    ctx = Context.get(P)
    rnd = random.Random(100 + shared + n)
    point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
    codec = Algorithm.FFT if use_omega else Algorithm.VANDERMONDE
    xs = [point(i).value for i in range(n)]
    ev = lambda poly, x: sum(co * pow(x, e, P) for e, co in enumerate(poly)) % P  # noqa: E731
    for liar_count in (t, t + 1):
        polys = structured_rows(rnd, P, c, t + 1)          # zero, constant, short, padded polynomials among the uniform ones
        liars = rnd.sample(range(n), liar_count)
        honest = [i for i in range(n) if i not in liars]
        rnd.shuffle(honest)
        fakes = []
        for poly in polys:
            q, deg = [rnd.randrange(1, P)] + [0] * t, 0
            for s_ in honest[:shared]:
                nq = [0] * (t + 1)
                for e in range(deg + 1):
                    nq[e + 1] = (nq[e + 1] + q[e]) % P
                    nq[e] = (nq[e] - q[e] * xs[s_]) % P
                q, deg = nq, deg + 1
            fakes.append([(a + b) % P for a, b in zip(poly, q)])
        cols = [[ev(fakes[j] if i in liars else polys[j], xs[i]) for j in range(c)] for i in range(n)]
        order = liars + honest
        host = IncrementalDecoder(EncoderFactory.get(point, codec), DecoderFactory.get(point, codec),
                                  RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO), degree=t, batch_size=c, max_errors=t)
        dev = DeviceIncrementalDecoder(P, n, t, batch_size=c, use_omega_powers=use_omega)
        for step, idx in enumerate(order):
            host.add(idx, cols[idx])
            dev.add(idx, ctx.upload_ints(cols[idx]))
            assert dev.done() == host.done(), (liar_count, step)
            assert dev._confirmed_errors == host._confirmed_errors and dev._z == host._z and dev._num_decoded == host._num_decoded, (liar_count, step)
            if host.done():
                break
        assert host.done() == dev.done()
        if host.done():
            hres, herr = host.get_results()
            dres, derr = dev.get_results()
            assert derr == herr
            assert ctx.download_ints(dres.reshape(-1, 4)) == [v for row in _padded(hres, t + 1) for v in row]
        if liar_count == t and shared == 0:
            assert dev.probes == 0 and dev.radius_verdicts >= 1          # the newest columns gave the true polynomial: nothing incremental ran


def test_device_decoder_randomised_vs_host_mirror():
    """~20 s, seeded: the bounded twin of scratch/stress_decoder.py (which found a missed Welch-Berlekamp refusal in the first
    version of the plan-free path: 94 divergences in 36 444 decodes, all of that one kind; 0 in 22 898 after the fix).  Random
    shapes / liar counts (up to one too many) / corruption patterns / arrival orders, both robust decoders, both point
    policies; after every column: same done / confirmed errors / arrival list / polynomials decoded, same exception type."""
    import time

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint
    from honeybadgermpc_amd.reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory

    rnd = random.Random(20260929)
    ctx = Context.get(P)
    t_end = time.time() + 20.0
    runs = robust_runs = raised = 0
    while time.time() < t_end or runs < 200:
        n = rnd.choice([10, 13, 16, 22, 31])
        t = rnd.randrange(3, (n - 1) // 3 + 1)
        c = rnd.choice([1, 2, 5, 17, 40])
        use_omega = rnd.random() < 0.3
        robust = rnd.choice(["gao", "gao", "wb"])
        point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
        xs = [point(i).value for i in range(n)]
        polys = structured_rows(rnd, P, c, t + 1)          # zero, constant, short, padded polynomials among the uniform ones
        cols = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
        liars = rnd.sample(range(n), rnd.randrange(0, t + 2))
        for i in liars:
            hit = {0: range(c), 1: [c - 1], 2: [rnd.randrange(c)], 3: rnd.sample(range(c), max(1, c // 3))}[rnd.randrange(4)]
            for j in hit:
                cols[i][j] = (cols[i][j] + 1 + rnd.randrange(P - 1)) % P
        order = list(range(n))
        rnd.shuffle(order)
        if rnd.random() < 0.5:
            order = liars + [i for i in order if i not in liars]
        codec = Algorithm.FFT if use_omega else Algorithm.VANDERMONDE
        host = IncrementalDecoder(EncoderFactory.get(point, codec), DecoderFactory.get(point, codec),
                                  RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO if robust == "gao" else Algorithm.WELCH_BERLEKAMP),
                                  degree=t, batch_size=c, max_errors=t)
        dev = DeviceIncrementalDecoder(P, n, t, batch_size=c, robust=robust, use_omega_powers=use_omega)
        what = (n, t, c, use_omega, robust, liars, order)
        for step, idx in enumerate(order):
            hexc = dexc = None
            try:
                host.add(idx, cols[idx])
            except BaseException as e:  # noqa: BLE001 - the reference re-raises bare Exceptions and assertion failures
                hexc = e
            try:
                dev.add(idx, ctx.upload_ints(cols[idx]))
            except BaseException as e:  # noqa: BLE001
                dexc = e
            assert type(hexc) is type(dexc), (what, step, repr(hexc), repr(dexc))
            if hexc is not None:
                raised += 1
                break
            assert dev.done() == host.done() and dev._confirmed_errors == host._confirmed_errors, (what, step)
            assert dev._z == host._z and dev._num_decoded == host._num_decoded, (what, step)
            if host.done():
                assert ctx.download_ints(dev.get_results()[0].reshape(-1, 4)) == [v for row in _padded(host.get_results()[0], t + 1) for v in row], what
                break
        robust_runs += dev.probes + dev.radius_verdicts + dev.launches > 0
        runs += 1
    assert runs >= 200 and robust_runs >= 50 and raised >= 5, (runs, robust_runs, raised)


def test_device_incremental_decoder_reference_transcripts(golden):
    """The transcripts tests/golden/incremental_decoder.json recorded from the reference's own IncrementalDecoder
    (done / result / confirmed errors after every add), replayed on the device decoder with the transcript's robust
    decoder (Gao -- batch_reconstruct's default, batch_reconstruction.py:85-90 -- or Welch-Berlekamp)."""
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder

    replayed = 0
    for tr in golden("incremental_decoder.json")["transcripts"]:
        ctx = Context.get(tr["p"])
        dec = DeviceIncrementalDecoder(tr["p"], tr["n"], tr["t"], batch_size=tr["batch"], use_omega_powers=tr["use_omega_powers"],
                                       robust="gao" if tr["robust"] == "gao" else "wb")
        for step in tr["steps"]:
            dec.add(step["idx"], tr["columns"][step["idx"]])
            res, errs = dec.get_results()
            assert dec.done() == step["done"]
            if step["result"] is None:
                assert res is None and errs is None
            else:
                got = ctx.download_ints(res.reshape(-1, ctx.n_limbs))
                d = tr["t"] + 1
                assert [got[i * d : (i + 1) * d] for i in range(tr["batch"])] == step["result"]
                assert sorted(errs) == step["errors"]
        assert dec.done()
        replayed += 1
    assert replayed >= 5


# ---- batch_reconstruct on device tensors, n parties in one process ---------------------------------------
class _Net:
    """queues standing in for the reference's router (router.py:66-107): send(dest, msg) / recv() -> (sender, msg)"""

    def __init__(self, n):
        self.q = [asyncio.Queue() for _ in range(n)]

    def send(self, i, tamper=None):
        def _send(dest, msg):
            self.q[dest].put_nowait((i, tamper(dest, msg) if tamper else msg))

        return _send

    def recv(self, i):
        return self.q[i].get


@pytest.mark.parametrize("n, t, b, use_omega, liars", [(4, 1, 7, False, 0), (4, 1, 7, False, 1), (7, 2, 50, False, 2), (7, 2, 31, True, 2),
                                                       (16, 5, 200, False, 5), (10, 3, 1, False, 3)])
def test_batch_reconstruct_device(n, t, b, use_omega, liars):
    """Every honest party opens the same secrets although `liars` parties send random columns in both rounds, one of
    them malformed bytes; the packed R1 messages of an honest party are the int lists the host batch_reconstruct sends."""
    import torch

    from honeybadgermpc_amd import wire
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device_reconstruction import batch_reconstruct_device
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint
    from honeybadgermpc_amd.reed_solomon import EncoderFactory

    rnd = random.Random(n * 1000 + b)
    ctx = Context.get(P)
    point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
    xs = [point(i).value for i in range(n)]
    secrets = [rnd.randrange(P) for _ in range(b)]
    polys = [[s] + [rnd.randrange(P) for _ in range(t)] for s in secrets]
    shares = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
    bad = set(rnd.sample(range(n), liars))
    sent = {}

    def tamper_for(i):
        def tamper(dest, msg):
            tag, blob = msg
            if i == min(bad) and tag == "R1" and dest % 2:
                return (tag, blob[:-3])                                   # truncated message
            count = wire.unpack_limbs(blob).shape[0]
            return (tag, wire.pack_ints([rnd.randrange(P) for _ in range(count)], P))

        def record(dest, msg):
            sent.setdefault((i, msg[0]), {})[dest] = msg[1]
            return msg

        return tamper if i in bad else record

    async def main():
        net = _Net(n)
        tasks = [batch_reconstruct_device(ctx.upload_ints(shares[i]), P, t, n, i, net.send(i, tamper_for(i)), net.recv(i), use_omega_powers=use_omega)
                 for i in range(n)]
        return await asyncio.gather(*tasks)

    results = asyncio.run(main())
    for i in range(n):
        if i not in bad:
            assert results[i] is not None and ctx.download_ints(results[i]) == secrets, i
    # the wire content of an honest party = what the host-level batch_reconstruct would have sent as int lists
    honest = min(set(range(n)) - bad)
    d = t + 1
    chunks = [shares[honest][k : k + d] + [0] * (d - len(shares[honest][k : k + d])) for k in range(0, b, d)]
    enc = EncoderFactory.get(point).encode(chunks)
    for dest in range(n):
        assert wire.unpack_ints(sent[(honest, "R1")][dest]) == [row[dest] for row in enc]
    torch.cuda.synchronize()


# ---- Mpc.open_share_array on the device: many concurrent small opens in one launch -------------------
class _TaggedNet:
    """the runtime's per-share-id channels (mpc.py:196-205): get_send_recv(tag) -> (send, recv) for party i"""

    def __init__(self, n):
        self.n, self.q = n, [dict() for _ in range(n)]

    def _queue(self, party, tag):
        return self.q[party].setdefault(tag, asyncio.Queue())

    def get_send_recv(self, i, tamper=None):
        def factory(tag):
            def send(dest, msg):
                self._queue(dest, tag).put_nowait((i, tamper(msg) if tamper else msg))

            return send, self._queue(i, tag).get

        return factory


@pytest.mark.parametrize("n, t, liars", [(4, 1, 0), (7, 2, 2), (16, 5, 1)])
def test_open_coalescer_many_small_opens(n, t, liars):
    """>= 64 concurrent opens of 1..40 shares each, issued like Mpc.open_share_array: one coalesced reconstruction per
    program step, results equal to opening each array on its own (= the secrets), also with liars sending garbage."""
    from honeybadgermpc_amd import wire
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.open_coalescer import OpenCoalescer

    rnd = random.Random(n * 31 + liars)
    ctx = Context.get(P)
    xs = list(range(1, n + 1))
    sizes = [rnd.randrange(1, 41) for _ in range(64)] + [0, 1] + [rnd.randrange(1, 41) for _ in range(6)]
    secrets = [[rnd.randrange(P) for _ in range(sz)] for sz in sizes]

    def share_all(vals):
        polys = [[s] + [rnd.randrange(P) for _ in range(t)] for s in vals]
        return [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]

    shares = [share_all(vals) for vals in secrets]          # [open][party][share]
    bad = set(rnd.sample(range(n), liars))

    def garble(msg):
        tag, blob = msg
        count = wire.unpack_limbs(blob).shape[0]
        return (tag, wire.pack_ints([rnd.randrange(P) for _ in range(count)], P))

    async def party(i, net, counters):
        co = OpenCoalescer(P, n, t, i, net.get_send_recv(i, garble if i in bad else None))
        # step 1: the first 66 opens are issued before anything is awaited -> one batch
        first = [co.open_share_array(ctx.upload_ints(shares[k][i]) if sizes[k] else ctx.empty(0)) for k in range(66)]
        got = [await h for h in first]
        # step 2: six more, awaited out of order -> a second batch
        second = [co.open_share_array(shares[k][i]) for k in range(66, 72)]
        got += [await h for h in reversed(second)][::-1]
        counters[i] = (co.opens, co.batches)
        assert co.pending_batches() == 0, "a finished batch is still held after its last open was delivered"
        assert (await first[3] is got[3])                  # a second await of a delivered open still answers
        return got

    async def main():
        net = _TaggedNet(n)
        counters = {}
        res = await asyncio.gather(*[party(i, net, counters) for i in range(n)])
        return res, counters

    results, counters = asyncio.run(main())
    for i in range(n):
        assert counters[i] == (72, 2), counters[i]
        if i in bad:
            continue
        for k in range(72):
            assert ctx.download_ints(results[i][k]) == secrets[k], (i, k)


def test_open_coalescer_cut_points_with_concurrent_coroutines():
    """with several coroutines per party the point where "the first await" falls depends on the scheduler, so
    (a) a batch cut by an await may only hold opens of ONE coroutine -- otherwise RuntimeError instead of parties exchanging
    batches of different composition; (b) flush() cuts at a point of the program order: two coroutines queue, a third one
    flushes, every party sees the same single batch whatever order its coroutines ran in; (c) nothing is kept once every
    open of a batch was delivered, also when some opens are never awaited."""
    import gc

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.open_coalescer import OpenCoalescer

    n, t = 4, 1
    rnd = random.Random(12)
    ctx = Context.get(P)
    xs = list(range(1, n + 1))
    secrets = [[rnd.randrange(P) for _ in range(5)] for _ in range(6)]

    def share_all(vals):
        polys = [[s] + [rnd.randrange(P) for _ in range(t)] for s in vals]
        return [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]

    shares = [share_all(v) for v in secrets]

    async def party(i, net, order):
        co = OpenCoalescer(P, n, t, i, net.get_send_recv(i), cut_on_await=False)
        handles = {}
        queued = asyncio.Event()

        turn = [0]

        async def producer(ks, delay):
            # the ORDER of open_share_array calls is the program order every party shares (the reference numbers share ids by
            # it); what differs from party to party is how the coroutines interleave around those calls
            for k in ks:
                while turn[0] != k:
                    await asyncio.sleep(0)
                for _ in range(delay):
                    await asyncio.sleep(0)
                handles[k] = co.open_share_array(shares[k][i])
                turn[0] += 1

        async def consumer(k):
            await queued.wait()
            return ctx.download_ints(await handles[k])

        prods = [asyncio.ensure_future(producer([0, 2, 4], order)), asyncio.ensure_future(producer([1, 3], 3 - order))]
        await asyncio.gather(*prods)
        with pytest.raises(RuntimeError, match="flush"):
            await handles[0]                              # cut_on_await=False: not cut yet -> loud
        co.flush()                                        # the deterministic cut: all five opens, one batch, on every party
        queued.set()
        got = await asyncio.gather(*[consumer(k) for k in (0, 1, 2, 3)])       # open 4 is never awaited
        assert co.batches == 1
        del handles
        gc.collect()
        assert co.pending_batches() == 0
        return got

    async def mixed(i, net):
        """cut_on_await=True and a batch holding opens of two coroutines: refused"""
        co = OpenCoalescer(P, n, t, i, net.get_send_recv(i))
        h = {}

        async def q(k):
            h[k] = co.open_share_array(shares[k][i])

        await asyncio.gather(q(0), q(1))
        with pytest.raises(RuntimeError, match="more than one coroutine"):
            await h[0]
        co.flush()
        return ctx.download_ints(await h[1])

    async def main():
        net = _TaggedNet(n)
        a = await asyncio.gather(*[party(i, net, i % 3) for i in range(n)])
        net2 = _TaggedNet(n)
        b = await asyncio.gather(*[mixed(i, net2) for i in range(n)])
        return a, b

    a, b = asyncio.run(main())
    for i in range(n):
        assert a[i] == secrets[:4] and b[i] == secrets[1]


def test_batch_reconstruct_device_survives_non_bytes_payloads():
    """a Byzantine sender behind an unpickling transport can send anything -- None, an int, a str, a list with
    a non-integer -- and must not abort an honest party's open; a well-formed list of ints (a reference-style party in a
    mixed deployment) is accepted as a column."""
    from honeybadgermpc_amd import wire
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device_reconstruction import batch_reconstruct_device

    n, t, b = 7, 2, 20
    rnd = random.Random(77)
    ctx = Context.get(P)
    xs = list(range(1, n + 1))
    secrets = [rnd.randrange(P) for _ in range(b)]
    polys = [[s] + [rnd.randrange(P) for _ in range(t)] for s in secrets]
    shares = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
    junk = {0: None, 1: 12345}

    def tamper_for(i):
        def tamper(dest, msg):
            tag, blob = msg
            if i in junk:
                return (tag, junk[i] if tag == "R1" else ["x", 1.5] if i == 0 else "R2?")
            if i == 2:                                   # honest, but speaks the reference's format: a list of Python ints
                return (tag, wire.unpack_ints(blob))
            if i == 3:                                   # honest values as non-canonical words (x + p < 2^256): residues count, as at
                import numpy as np                       # the reference's boundary (to_ZZ_p, pyx:31-32)

                vals = [v + P if v + P < (1 << 256) else v for v in wire.unpack_ints(blob)]
                raw = np.array([[(v >> (64 * q)) & ((1 << 64) - 1) for q in range(4)] for v in vals], dtype=np.uint64)
                return (tag, wire.pack_limbs(raw))
            return msg

        return tamper

    async def main():
        net = _Net(n)
        tasks = [batch_reconstruct_device(ctx.upload_ints(shares[i]), P, t, n, i, net.send(i, tamper_for(i)), net.recv(i)) for i in range(n)]
        return await asyncio.gather(*tasks)

    results = asyncio.run(main())
    for i in range(2, n):
        assert results[i] is not None and ctx.download_ints(results[i]) == secrets, i


def test_robust_reconstruct_device_single_share():
    """robust_reconstruction.py:14-30 on the device decoder: batch size 1, t liars among n = 3t + 1"""
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.open_coalescer import robust_reconstruct_device

    n, t = 10, 3
    rnd = random.Random(5)
    ctx = Context.get(P)
    poly = [rnd.randrange(P) for _ in range(t + 1)]
    vals = [sum(co * pow(i + 1, e, P) for e, co in enumerate(poly)) % P for i in range(n)]
    liars = {1, 4, 8}
    for i in liars:
        vals[i] = rnd.randrange(P)

    async def main():
        loop = asyncio.get_event_loop()
        futs = []
        for i in [4, 1, 0, 2, 8, 3, 5, 6, 7, 9]:           # liars arrive early
            f = loop.create_future()
            f.set_result(vals[i])
            futs.append((i, f))
        ordered = [None] * n
        for i, f in futs:
            ordered[i] = f
        return await robust_reconstruct_device(ordered, P, n, t)

    coeffs, errors = asyncio.run(main())
    assert coeffs is not None and ctx.download_ints(coeffs) == poly and errors <= liars
