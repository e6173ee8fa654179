"""
Host-side logic of the drop-in package against the reference's behaviour:
  * the known-answer vectors of the reference's codec / reconstruction tests
    (/root/reference/tests/test_reed_solomon.py, test_reed_solomon_wb.py,
     test_batch_reconstruction.py, test_polynomial.py, test_field.py), restated;
  * golden transcripts of the reference's IncrementalDecoder and batch_reconstruct
    (tests/golden/incremental_decoder.json, batch_reconstruct.json).
Every test runs twice: backend='oracle' (CPU oracle injected by the test harness, runs
anywhere) and backend='hip' (the real product path, gpu-marked).
"""
import asyncio
import random
from unittest.mock import patch

import pytest

from conftest import BLS
from honeybadgermpc_amd.field import GF, FieldsNotIdentical, GFElement
from honeybadgermpc_amd.polynomial import EvalPoint, fnt_decode_step1, fnt_decode_step2, get_omega, polynomials_over
from honeybadgermpc_amd.utils.misc import chunk_data, flatten_lists, transpose_lists, wrap_send


# ------------------------------------------------------------------ pure host pieces (no backend)
def test_misc_golden(golden):
    g = golden("misc.json")
    for c in g["chunk_data"]:
        assert chunk_data(list(c["data"]), c["size"]) == c["out"]
    for c in g["transpose_lists"]:
        assert transpose_lists(c["in"]) == c["out"]
    for c in g["flatten_lists"]:
        assert flatten_lists(c["in"]) == c["out"]
    assert chunk_data([], 2) == [0, 0]  # the reference's flat-list quirk (utils/misc.py:40-41)
    with pytest.raises(TypeError):
        chunk_data((1, 2), 2)
    sent = []
    wrap_send("T", lambda d, m: sent.append((d, m)))(3, "x")
    assert sent == [(3, ("T", "x"))]


def test_field():  # reference tests/test_field.py
    f, f2 = GF(BLS), GF(BLS)
    assert f is f2
    assert GF(53) is not f
    with pytest.raises(ValueError):
        GF(15)
    a, b = f(5), f(7)
    assert (a + b) == 12 and (a - b) == f(-2) and (a * b) == 35 and (a / b) * b == a
    assert (~a) * a == 1 and a ** 3 == 125 and -a == f(BLS - 5)
    assert 3 + a == 8 and 3 - a == f(-2) and 3 * a == 15 and (3 / a) * a == 3
    with pytest.raises(FieldsNotIdentical):
        a + GF(53)(1)
    with pytest.raises(ZeroDivisionError):
        ~f(0)
    assert type(a + 1) is GFElement and bool(f(0)) is False and hash(f(3)) == hash(f(3))
    for fld in (f, GF(53), GF(13)):
        rnd = random.Random(3)
        for _ in range(20):
            sq = fld(rnd.randrange(1, fld.modulus)) ** 2  # sqrt(0) asserts, as in the reference (field.py:175)
            assert sq.sqrt() ** 2 == sq
    assert f.random(0).value == 8063396892870388055806370369789704857755116044327394765020751373651916505604


def test_constants(golden):
    c = golden("constants.json")
    f = GF(BLS)
    assert f.random(0).value == c["seed0_random"]
    for order, w in c["omega"].items():
        assert get_omega(f, int(order), seed=0).value == w
    for ep in c["evalpoint"]:
        pt = EvalPoint(f, ep["n"], use_omega_powers=True)
        assert (pt.order, pt.omega.value, pt.omega2.value) == (ep["order"], ep["omega"], ep["omega2"])
        assert [pt(i).value for i in range(ep["n"])] == ep["points"]
    pt = EvalPoint(f, 5)
    assert [pt(i).value for i in range(5)] == [1, 2, 3, 4, 5] and pt.omega is None and pt.order == 5
    assert pt.zero() == 0


def test_polynomial_python(galois_field, polynomial):  # reference tests/test_polynomial.py:6-70,72-92
    poly = polynomial
    p1 = poly([1, 5, 3, 15, 0, 3])
    assert p1(3) == 1 + 15 + 27 + 405 + 0 + 729 and p1.degree() == 5
    assert poly([1, 2, 0, 0]).coeffs == [1, 2] and poly([0]).is_zero() and poly([]).coeffs == []
    q, r = divmod(poly([1, 2, 1]), poly([1, 1]))
    assert q == poly([1, 1]) and r.is_zero()
    rnd = random.Random(5)
    xs = [galois_field(rnd.randrange(BLS)) for _ in range(6)]
    f = poly([rnd.randrange(BLS) for _ in range(6)])
    pts = [(x, f(x)) for x in xs]
    assert poly.interpolate(pts) == f
    assert poly.interpolate_at(pts, 0) == f(0)
    n = 8
    omega = get_omega(galois_field, n, seed=1)
    ev = f.evaluate_fft(omega, n)
    assert ev == [f(omega ** i) for i in range(n)]
    assert poly.interpolate_fft(ev, omega) == f
    # Python FNT decode round trip (reference tests/test_polynomial.py:72-92)
    d, nn = 10, 16
    omega2 = get_omega(galois_field, 2 * nn, seed=1)
    g = poly([rnd.randrange(BLS) for _ in range(d)])
    zs = rnd.sample(range(nn), d)
    ys = [g((omega2 ** 2) ** z) for z in zs]
    as_, ais_ = fnt_decode_step1(poly, zs, omega2, nn)
    assert fnt_decode_step2(poly, zs, ys, as_, ais_, omega2, nn) == g


# ------------------------------------------------------------------ codec classes over a backend
def test_polynomial_native(backend, galois_field, polynomial):  # test_polynomial.py:94-107 (interp_extrap_cpp)
    rnd = random.Random(6)
    n = 8
    omega = get_omega(galois_field, 2 * n, seed=2)
    xs = [rnd.randrange(BLS) for _ in range(n)]
    want = [v.value for v in polynomial.interp_extrap([galois_field(x) for x in xs], omega)]
    assert polynomial.interp_extrap_cpp(xs, omega) == want


def test_codecs_reference_vectors(backend, galois_field):  # tests/test_reed_solomon.py:19-183
    from honeybadgermpc_amd import reed_solomon as rs

    p = galois_field.modulus
    pt = EvalPoint(galois_field, 4)
    ptw = EvalPoint(galois_field, 4, use_omega_powers=True)
    w = ptw.omega.value
    for enc in (rs.VandermondeEncoder(pt), rs.EncoderFactory.get(pt)):
        assert enc.encode([1, 2]) == [3, 5, 7, 9]
        assert enc.encode([[1, 2], [2, 3]]) == [[3, 5, 7, 9], [5, 8, 11, 14]]
        assert enc.encode(((1, 2), (2, 3))) == [[3, 5, 7, 9], [5, 8, 11, 14]]  # tuples dispatch as batch
    fft_expected = [(2 * pow(w, i, p) + 1) % p for i in range(4)]
    for enc in (rs.FFTEncoder(ptw), rs.EncoderFactory.get(ptw)):
        assert enc.encode([1, 2]) == fft_expected
        assert enc.encode([[1, 2]]) == [fft_expected]
    for dec in (rs.VandermondeDecoder(pt), rs.DecoderFactory.get(pt)):
        assert dec.decode([1, 3], [5, 9]) == [1, 2]
        assert dec.decode([1, 3], [[5, 9], [8, 14]]) == [[1, 2], [2, 3]]
    for dec in (rs.FFTDecoder(ptw), rs.DecoderFactory.get(ptw)):
        assert dec.decode([1, 3], [fft_expected[1], fft_expected[3]]) == [1, 2]
    for mk in (rs.GaoRobustDecoder, rs.WelchBerlekampRobustDecoder):
        assert mk(1, pt).robust_decode([0, 1, 2, 3], [3, 5, 0, 9]) == ([1, 2], [2])
        bad = list(fft_expected)
        bad[2] = 0
        assert mk(1, ptw).robust_decode([0, 1, 2, 3], bad) == ([1, 2], [2])
    with pytest.raises(ValueError):
        rs.EncoderFactory.get(pt, "nope")
    with pytest.raises(ValueError):
        rs.RobustDecoderFactory.get(1, pt, "nope")
    assert isinstance(rs.RobustDecoderFactory.get(1, pt), rs.GaoRobustDecoder)
    assert isinstance(rs.RobustDecoderFactory.get(1, pt, rs.Algorithm.WELCH_BERLEKAMP), rs.WelchBerlekampRobustDecoder)


def test_selector_policy(galois_field):  # tests/test_reed_solomon.py:186-277 (pure policy, no arithmetic)
    from honeybadgermpc_amd import reed_solomon as rs
    from honeybadgermpc_amd.ntl import AvailableNTLThreads

    def pt(n):
        return EvalPoint(galois_field, n, use_omega_powers=True)

    for n, cls in [(4, rs.VandermondeEncoder), (65, rs.VandermondeEncoder), (40, rs.VandermondeEncoder),
                   (120, rs.FFTEncoder), (55, rs.FFTEncoder), (255, rs.FFTEncoder), (257, rs.FFTEncoder)]:
        for k in (1, 100000):
            assert isinstance(rs.EncoderSelector.select(pt(n), k), cls)
    with patch("psutil.cpu_count") as cpu:
        for cores in (1, 100):
            cpu.return_value = cores
            for b in (1, 1000, 100000):
                rs.DecoderSelector.set_optimal_thread_count(b)
                assert isinstance(rs.DecoderSelector.select(pt(4), b), rs.VandermondeDecoder)
        for cores in (1, 2, 4, 8):
            cpu.return_value = cores
            for b in (1, 16, 32):
                rs.DecoderSelector.set_optimal_thread_count(b)
                assert isinstance(rs.DecoderSelector.select(pt(65), b), rs.FFTDecoder)
            for b in (512, 1024, 2048, 4096):
                rs.DecoderSelector.set_optimal_thread_count(b)
                assert isinstance(rs.DecoderSelector.select(pt(65), b), rs.VandermondeDecoder)
        for n in (32, 64, 128, 256):
            for cores in (1, 2, 4, 8, 16):
                cpu.return_value = cores
                for b in [2 ** i for i in range(16)]:
                    rs.DecoderSelector.set_optimal_thread_count(b)
                    want = rs.VandermondeDecoder if b > 0.5 * n * min(b, AvailableNTLThreads()) else rs.FFTDecoder
                    assert isinstance(rs.DecoderSelector.select(pt(n), b), want)


def test_wb_module(backend):  # tests/test_reed_solomon_wb.py:6-75
    from honeybadgermpc_amd.reed_solomon_wb import make_wb_encoder_decoder

    rnd = random.Random(8)
    k, n, p = 8, 22, 53
    t = k - 1
    cmax, emax = n - 2 * t - 1, (n - 2 * t - 1) // 2

    def corrupt(message, ne, nn):
        message = list(message)
        idx = rnd.sample(range(len(message)), ne + nn)
        for i in range(ne):
            message[idx[i]] = rnd.randint(0, 131)
        for i in range(nn):
            message[idx[i + ne]] = None
        return message

    for int_msg, want in [([2, 3, 2, 8, 7, 5, 9, 5], [2, 3, 2, 8, 7, 5, 9, 5]), ([0] * 8, [])]:
        enc, dec, solve = make_wb_encoder_decoder(n, k, p)
        encoded = enc(int_msg)
        for ne, nn in [(0, 0), (0, cmax), (emax, 0), (emax // 2, cmax // 4)]:
            assert dec(corrupt(encoded, ne, nn), debug=False) == want
    with pytest.raises(Exception):
        make_wb_encoder_decoder(60, 8, 53)
    # solve_system (host helper): E | Q and Q / E is the message
    enc, dec, solve = make_wb_encoder_decoder(10, 3, 53)
    from honeybadgermpc_amd.polynomial import EvalPoint as EP

    f = GF(53)
    word = enc([4, 5, 6])
    word[2] = word[2] + 1
    pt = EP(f, 10)
    q, e = solve([(pt(i), w) for i, w in enumerate(word)], max_e=3)
    quo, rem = divmod(q, e)
    assert rem.is_zero() and [c.value for c in quo.coeffs] == [4, 5, 6]


def test_wb_golden_through_decoder(backend, golden):
    """reference WB outputs (incl. failure messages) through make_wb_encoder_decoder.decode"""
    from honeybadgermpc_amd.reed_solomon_wb import make_wb_encoder_decoder

    cases = golden("welch_berlekamp.json")["cases"]
    decs = {}
    for case in cases[::3]:
        key = (case["n"], case["k"], case["p"])
        if key not in decs:
            decs[key] = make_wb_encoder_decoder(*key)[1]
        word = case["word"]
        if case["error"] is None:
            assert [c.value for c in decs[key](word, debug=False)] == case["coeffs"]
        else:
            with pytest.raises(Exception) as ei:
                decs[key](word, debug=False)
            assert str(ei.value) == case["error"]


# ------------------------------------------------------------------ IncrementalDecoder transcripts
def test_incremental_decoder_transcripts(backend, golden):
    from honeybadgermpc_amd import reed_solomon as rs

    for tr in golden("incremental_decoder.json")["transcripts"]:
        fp = GF(tr["p"])
        point = EvalPoint(fp, tr["n"], use_omega_powers=tr["use_omega_powers"])
        algo = rs.Algorithm.FFT if tr["use_omega_powers"] else rs.Algorithm.VANDERMONDE
        enc, dec = rs.EncoderFactory.get(point, algo), rs.DecoderFactory.get(point, algo)
        rdec = rs.RobustDecoderFactory.get(tr["t"], point, algorithm=tr["robust"])
        inc = rs.IncrementalDecoder(enc, dec, rdec, degree=tr["t"], batch_size=tr["batch"], max_errors=tr["t"])
        for step in tr["steps"]:
            inc.add(step["idx"], tr["columns"][step["idx"]])
            res, errs = inc.get_results()
            assert inc.done() == step["done"]
            assert res == step["result"]
            assert (None if errs is None else sorted(errs)) == step["errors"]
        assert inc.done()
        assert inc.get_results()[0] == tr["msgs"]
    # validation errors and duplicate senders (reference reed_solomon.py:288-300, 369-372)
    fp = GF(BLS)
    point = EvalPoint(fp, 4)
    inc = rs.IncrementalDecoder(rs.VandermondeEncoder(point), rs.VandermondeDecoder(point), rs.GaoRobustDecoder(1, point), 1, 2, 1)
    with pytest.raises(rs.DecodeValidationError):
        inc.add(0, [1])
    assert inc.get_results() == (None, None)


# ------------------------------------------------------------------ batch_reconstruct
class _Router:
    """n-party in-process message router (the reference's SimpleRouter, router.py:66-107, in spirit)."""

    def __init__(self, n):
        self.queues = [asyncio.Queue() for _ in range(n)]
        self.sent = [{"R1": [None] * n, "R2": None} for _ in range(n)]

    def send(self, i):
        def _send(dest, msg):
            tag, payload = msg
            if tag == "R1":
                self.sent[i]["R1"][dest] = list(payload)
            else:
                self.sent[i]["R2"] = list(payload)
            self.queues[dest].put_nowait((i, msg))

        return _send

    def recv(self, i):
        return self.queues[i].get


class _Cfg:
    def __init__(self, algo):
        self.induce_faults = False
        self.decoding_algorithm = algo


def _run_batch(p, t, n, shares, use_omega, robust, skip=()):
    from honeybadgermpc_amd.batch_reconstruction import batch_reconstruct

    fp = GF(p)

    async def go():
        router = _Router(n)
        tasks = [
            batch_reconstruct([fp(v) for v in shares[i]], p, t, n, i, router.send(i), router.recv(i),
                              config=_Cfg(robust), use_omega_powers=use_omega)
            for i in range(n) if i not in skip
        ]
        return await asyncio.gather(*tasks), router

    return asyncio.run(go())


def test_batch_reconstruct_reference_vectors(backend):  # tests/test_batch_reconstruction.py:11-194
    p, n, t = BLS, 4, 1
    shares = [(3, 7, 4), (4, 10, 6), (5, 13, 8), (6, 16, 10)]  # x+2, 3x+4, 2x+2 at x=1..4
    results, _ = _run_batch(p, t, n, shares, False, "gao")
    for r in results:
        assert all(type(e) is GFElement for e in r) and r == [2, 4, 2]
    bad = [list(s) for s in shares]
    bad[1] = [0, 0, 0]
    results, _ = _run_batch(p, t, n, bad, False, "gao")
    assert all(r == [2, 4, 2] for r in results)
    w = EvalPoint(GF(p), n, use_omega_powers=True).omega.value
    fshares = [[(pow(w, i, p) + 2) % p, (3 * pow(w, i, p) + 4) % p] for i in range(n)]
    results, _ = _run_batch(p, t, n, fshares, True, "gao")
    assert all(r == [2, 4] for r in results)
    fshares[1] = [0, 0]
    results, _ = _run_batch(p, t, n, fshares, True, "gao")
    assert all(r == [2, 4] for r in results)


def test_batch_reconstruct_timeout(backend):  # tests/test_batch_reconstruction.py:113-132
    from honeybadgermpc_amd.batch_reconstruction import batch_reconstruct

    p, n, t = BLS, 4, 1
    fp = GF(p)
    shares = [(3, 7, 4), (0, 0, 0), (5, 13, 8), (6, 16, 10)]

    async def go():
        router = _Router(n)
        tasks = [batch_reconstruct([fp(v) for v in shares[i]], p, t, n, i, router.send(i), router.recv(i))
                 for i in range(n) if i != 2]
        await asyncio.wait_for(asyncio.gather(*tasks), timeout=1)

    with pytest.raises(asyncio.TimeoutError):
        asyncio.run(go())


def test_batch_reconstruct_golden_runs(backend, golden):
    for run in golden("batch_reconstruct.json")["runs"]:
        results, router = _run_batch(run["p"], run["t"], run["n"], run["shares"], run["use_omega_powers"], run["robust"])
        outs = [None if r is None else [v.value for v in r] for r in results]
        assert outs == run["outputs"]
        for i in range(run["n"]):
            assert router.sent[i]["R1"] == run["sent"][i]["R1"]
            assert router.sent[i]["R2"] == run["sent"][i]["R2"]
        assert all(o == run["secrets"] for o in outs)


def test_robust_reconstruct(backend):
    from honeybadgermpc_amd.robust_reconstruction import robust_reconstruct

    p, n, t = BLS, 7, 2
    fp = GF(p)
    poly = polynomials_over(fp)
    point = EvalPoint(fp, n)
    f = poly([42, 7, 9])
    vals = [f(point(i)) for i in range(n)]
    vals[3] = fp(1)

    async def go():
        loop = asyncio.get_event_loop()
        futs = []
        for v in vals:
            fut = loop.create_future()
            fut.set_result(v)
            futs.append(fut)
        return await robust_reconstruct(futs, fp, n, t, point, t)

    got, errors = asyncio.run(go())
    # completion order of already-resolved futures is a set order: party 3 may or may not have been
    # consumed before 2t+1 columns agreed (the reference behaves the same way)
    assert got == f and set(errors) <= {3}


# ------------------------------------------------------------------ batched robust path
@pytest.mark.parametrize("robust", ["gao", "welch-berlekamp"])
@pytest.mark.parametrize("use_omega", [False, True])
def test_batched_robust_update_equals_sequential(backend, robust, use_omega):
    """IncrementalDecoder with robust_decode_batch must be indistinguishable from the reference's
    one-polynomial-at-a-time loop: same done() trajectory, results and confirmed-error sets, for
    faulty parties that corrupt all, some, or a single polynomial of the batch."""
    from honeybadgermpc_amd import reed_solomon as rs

    if robust == "welch-berlekamp" and use_omega:
        pytest.skip("same code path as Vandermonde points")
    rnd = random.Random(17 + use_omega)
    p, n, t, batch = BLS, 13, 4, 9
    fp = GF(p)
    point = EvalPoint(fp, n, use_omega_powers=use_omega)
    algo = rs.Algorithm.FFT if use_omega else rs.Algorithm.VANDERMONDE
    enc, dec = rs.EncoderFactory.get(point, algo), rs.DecoderFactory.get(point, algo)
    for trial in range(6):
        msgs = [[rnd.randrange(p) for _ in range(t + 1)] for _ in range(batch)]
        encoded = enc.encode(msgs)
        columns = [[encoded[b][j] for b in range(batch)] for j in range(n)]
        bad = rnd.sample(range(n), rnd.randrange(1, t + 1))
        for j in bad:
            style = rnd.randrange(3)
            rows = range(batch) if style == 0 else rnd.sample(range(batch), 1 if style == 1 else batch // 2)
            for b in rows:
                columns[j][b] = rnd.randrange(p)
        order = list(range(n))
        rnd.shuffle(order)

        class Sequential:
            def __init__(self, inner):
                self.robust_decode = inner.robust_decode     # no robust_decode_batch attribute

        traces = []
        for wrap in (False, True):
            rdec = rs.RobustDecoderFactory.get(t, point, algorithm=robust)
            if wrap:
                rdec = Sequential(rdec)
            inc = rs.IncrementalDecoder(enc, dec, rdec, degree=t, batch_size=batch, max_errors=t)
            trace = []
            for idx in order:
                try:
                    inc.add(idx, list(columns[idx]))
                except Exception as e:  # noqa: BLE001
                    # the reference's WB path asserts 2t+1+c <= n once confirmed errors shrink the
                    # arrival set (reed_solomon_wb.py:132); both modes must fail identically
                    trace.append((idx, "raise", type(e).__name__, str(e)))
                    break
                res, errs = inc.get_results()
                trace.append((idx, inc.done(), res, None if errs is None else sorted(errs), inc._num_decoded, list(inc._z)))
                if inc.done():
                    break
            traces.append(trace)
            if trace[-1][1] is True:
                assert inc.get_results()[0] == msgs
        assert traces[0] == traces[1]
        assert robust != "gao" or traces[0][-1][1] is True


def test_wire_format_roundtrip():
    import numpy as np

    from honeybadgermpc_amd import wire

    rnd = random.Random(1)
    vals = [rnd.randrange(BLS) for _ in range(37)] + [0, BLS - 1]
    blob = wire.pack_ints(vals, BLS)
    assert len(blob) == 16 + 32 * len(vals) and blob[:4] == b"HBFE"
    assert wire.unpack_ints(blob) == vals
    assert wire.unpack_ints(wire.pack_ints([BLS + 5], BLS)) == [5]
    a = wire.unpack_limbs(blob)
    assert a.shape == (len(vals), 4) and wire.pack_limbs(a) == blob
    t = wire.wire_to_tensor(blob)
    assert wire.tensor_to_wire(t) == blob
    for bad in (blob[:10], b"XXXX" + blob[4:], blob + b"\0"):
        with pytest.raises(ValueError):
            wire.unpack_limbs(bad)
    assert wire.unpack_ints(wire.pack_limbs(np.zeros((0, 4), dtype=np.uint64))) == []


@pytest.mark.parametrize("n,t,k", [(4, 1, 3), (4, 1, 4), (7, 2, 5), (16, 5, 13)])
def test_random_refinement(backend, n, t, k):  # reference tests/progs/test_random_refinement.py
    """refine_randoms is linear, so refined share vectors of the n parties must again be
    consistent degree-t sharings: every subset of t+1 parties reconstructs the same values,
    and those values are the refinement of the secrets."""
    from honeybadgermpc_amd.progs.random_refinement import refine_randoms

    rnd = random.Random(n * 100 + k)
    fp = GF(BLS)
    poly = polynomials_over(fp)
    point = EvalPoint(fp, n)
    secrets = [rnd.randrange(BLS) for _ in range(k)]
    sharings = [poly([s] + [rnd.randrange(BLS) for _ in range(t)]) for s in secrets]
    per_party = [[sharings[j](point(i)).value for j in range(k)] for i in range(n)]
    refined = [refine_randoms(n, t, fp, per_party[i]) for i in range(n)]
    assert all(len(r) == k - t for r in refined)
    want = refine_randoms(n, t, fp, secrets)
    for subset in ([0, 1, 2, 3, 4, 5][: t + 1], list(range(n - t - 1, n))):
        for j in range(k - t):
            got = poly.interpolate_at([(point(i), fp(refined[i][j])) for i in subset], 0)
            assert got == want[j]


def test_share_files_text_and_packed(tmp_path):
    """reference preprocessing.py:106-169: the text format (round trip, append rule, metadata checks) and its packed twin."""
    import random

    from honeybadgermpc_amd import wire
    from honeybadgermpc_amd._capi import ints_to_limbs, limbs_to_ints

    rnd = random.Random(3)
    vals = [rnd.randrange(BLS) for _ in range(50)] + [0, 1, BLS - 1]
    name = wire.share_filename(str(tmp_path / "triples"), 4, 1, 2)
    assert name.endswith("triples_4_1-2.share")
    wire.write_share_file(name, BLS, 1, 2, vals[:20])
    lines = open(name).read().splitlines()
    assert lines[:3] == [str(BLS), "1", "2"] and [int(v) for v in lines[3:]] == vals[:20]      # the reference's layout, line by line
    wire.write_share_file(name, BLS, 1, 2, vals[20:], append=True)
    assert wire.read_share_file(name, BLS) == (1, 2, vals)
    with pytest.raises(AssertionError):
        wire.write_share_file(name, BLS, 2, 2, [1], append=True)                                # different degree
    with pytest.raises(AssertionError):
        wire.read_share_file(name, 13)
    wire.write_share_file(name, BLS, 1, 2, vals[:5])                                            # overwrite
    assert wire.read_share_file(name, BLS) == (1, 2, vals[:5])
    fresh = str(tmp_path / "fresh_4_1-0.share")
    wire.write_share_file(fresh, BLS, 1, 0, vals[:3], append=True)                              # append to a missing file creates it
    assert wire.read_share_file(fresh, BLS) == (1, 0, vals[:3])

    packed = str(tmp_path / "rands_4_1-3.shareb")
    limbs = ints_to_limbs(vals, BLS, 32)
    wire.write_share_file_packed(packed, BLS, 1, 3, limbs[:10])
    wire.write_share_file_packed(packed, BLS, 1, 3, limbs[10:], append=True)
    degree, ctx_id, got = wire.read_share_file_packed(packed, BLS)
    assert (degree, ctx_id) == (1, 3) and limbs_to_ints(got, 32) == vals
    with pytest.raises(AssertionError):
        wire.write_share_file_packed(packed, BLS, 1, 1, limbs[:1], append=True)                 # different party
    with pytest.raises(AssertionError):
        wire.read_share_file_packed(packed, (1 << 255) - 19)
    with open(packed, "ab") as f:
        f.write(b"\x00" * 5)
    with pytest.raises(ValueError):
        wire.read_share_file_packed(packed, BLS)
