"""
GPU parity: the HIP path (through the C ABI / the ntl drop-in) against the CPU oracle on the
same seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full
sizes -- through size-independent properties (encode -> erase -> decode round trips,
linearity, consistency of validation).  Bit-exact everywhere: this is integer work.
"""
import ctypes
import os
import random

import numpy as np
import pytest

import oracle
from conftest import BLS, clear_hook, set_hook

pytestmark = pytest.mark.gpu

P = BLS
PRIMES = [P, 13, 53, (1 << 256) - 189, (1 << 255) - 19, (1 << 64) - 59]


@pytest.fixture(scope="module")
def hip():
    import torch

    assert torch.cuda.is_available()
    from honeybadgermpc_amd import ntl

    return ntl


def rand_rows(rnd, p, c, d):
    """rows of coefficients for every test of this file: zero, constant, short and padded rows among the uniform ones (tests/structured.py)"""
    from structured import structured_rows

    return structured_rows(rnd, p, c, d)


# ------------------------------------------------------------------ C ABI sanity
def test_capi_context_errors():
    from honeybadgermpc_amd._capi import HB_ERR_BAD_ARG, HB_ERR_UNSUPPORTED, ints_to_limbs, load_library, np_ptr

    lib = load_library()     # imports torch first: one HIP runtime per process (see _capi.load_library)
    assert lib.hb_device_count() >= 1
    h = ctypes.c_void_p()
    even = ints_to_limbs([10], 11)
    assert lib.hb_ctx_create(ctypes.byref(h), np_ptr(even), 4, 0) == HB_ERR_UNSUPPORTED
    assert lib.hb_ctx_create(ctypes.byref(h), np_ptr(even), 3, 0) == HB_ERR_BAD_ARG
    odd = ints_to_limbs([13], 14)
    assert lib.hb_ctx_create(ctypes.byref(h), np_ptr(odd), 4, 0) == 0
    assert lib.hb_elem_bytes(h) == 32
    lib.hb_ctx_destroy(h)


def test_narrow_context_matches_wide():
    """the 1-limb (p < 2^64) instantiation of every kernel against the 4-limb one"""
    import torch

    from honeybadgermpc_amd._capi import ints_to_limbs, limbs_to_ints, load_library, np_ptr

    lib = load_library()
    rnd = random.Random(21)
    for p in (13, 53, (1 << 64) - 59, 0xFFFFFFFF00000001):
        n, d, c = 20, 7, 150
        x = [rnd.randrange(1, p) for _ in range(n)] if p > 100 else list(range(1, min(n, p - 1) + 1))
        n = len(x)
        polys = rand_rows(rnd, p, c, d)
        want = oracle.vandermonde_batch_evaluate(x, polys, p)
        h = ctypes.c_void_p()
        assert lib.hb_ctx_create(ctypes.byref(h), np_ptr(ints_to_limbs([p], p + 1, 8)), 1, 0) == 0
        assert lib.hb_elem_bytes(h) == 8
        din = torch.from_numpy(ints_to_limbs([v for r in polys for v in r], p, 8).view(np.int64).copy()).cuda()
        dout = torch.empty((c * n, 1), dtype=torch.int64, device="cuda")
        rc = lib.hb_vandermonde_batch_evaluate(h, np_ptr(ints_to_limbs(x, p, 8)), n, ctypes.c_void_p(din.data_ptr()), c, d,
                                               ctypes.c_void_p(dout.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        got = limbs_to_ints(dout.cpu().numpy().view(np.uint64), 8)
        assert [got[i * n : (i + 1) * n] for i in range(c)] == want
        dk = min(d, n)
        ys = torch.from_numpy(ints_to_limbs([v for r in want for v in r[:dk]], p, 8).view(np.int64).copy()).cuda()
        dec = torch.empty((c * dk, 1), dtype=torch.int64, device="cuda")
        rc = lib.hb_vandermonde_batch_interpolate(h, np_ptr(ints_to_limbs(x[:dk], p, 8)), dk, ctypes.c_void_p(ys.data_ptr()), c,
                                                  ctypes.c_void_p(dec.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        got = limbs_to_ints(dec.cpu().numpy().view(np.uint64), 8)
        if dk == d:
            assert [got[i * d : (i + 1) * d] for i in range(c)] == polys
        lib.hb_ctx_destroy(h)


# ------------------------------------------------------------------ Vandermonde path
@pytest.mark.parametrize("p", PRIMES)
def test_vandermonde_vs_oracle(hip, p):
    rnd = random.Random(hash(p) & 0xFFFF)
    for n, d, c in [(4, 2, 256), (16, 6, 70), (64, 22, 130), (100, 34, 9), (7, 7, 64), (1, 1, 1), (65, 3, 5), (33, 33, 3)]:
        if n >= p:
            n = p - 1
            d = min(d, n)
        x = list(range(1, n + 1))
        polys = rand_rows(rnd, p, c, d)
        polys[0] = [0] * d
        polys[-1] = [p - 1] * d
        want = oracle.vandermonde_batch_evaluate(x, polys, p)
        assert hip.vandermonde_batch_evaluate(x, polys, p) == want
        z = rnd.sample(range(n), d)
        xz = [x[i] for i in z]
        ys = [[row[i] for i in z] for row in want]
        assert hip.vandermonde_batch_interpolate(xz, ys, p) == polys
        assert hip.vandermonde_batch_interpolate(xz, ys, p) == oracle.vandermonde_batch_interpolate(xz, ys, p)


def test_vandermonde_points_with_and_without_high_digits_in_the_first_terms(hip, monkeypatch):
    """k_mm8 leaves the second digit group of K-block 0 out when no entry x^l, l < 8, has a digit above the eighth (the points 1 .. n);
    points around 1000 do have them (1000^7 > 2^69) while their powers still fit the small-entry kernel (1000^11 < 2^110): both
    variants against the oracle, and the skipping one against itself with the skip switched off"""
    rnd = random.Random(77)
    for x, d, c in [(list(range(990, 1006)), 12, 300), ([3, 900, 1000, 17, 256, 255, 257, 1], 10, 270), (list(range(1, 65)), 22, 300), (list(range(1, 33)), 11, 260)]:
        polys = rand_rows(rnd, P, c, d)
        want = oracle.vandermonde_batch_evaluate(x, polys, P)
        assert hip.vandermonde_batch_evaluate(x, polys, P) == want
    set_hook(monkeypatch, "HB_MM8_NO_SKIP", "1")
    x, d, c = list(range(2, 66)), 22, 300          # a point set no test has used: a fresh image, built with the switch set
    polys = rand_rows(rnd, P, c, d)
    assert hip.vandermonde_batch_evaluate(x, polys, P) == oracle.vandermonde_batch_evaluate(x, polys, P)


@pytest.mark.parametrize("p", [P, (1 << 256) - 189, (1 << 64) - 59, 257])
def test_vandermonde_a_few_polynomials(hip, monkeypatch, p):
    """up to eight polynomials of eight or more coefficients (what the device decoder evaluates: ONE candidate at all parties' points) take
    k_eval_few -- a workgroup per (polynomial, point), powers by square and multiply, a tree of additions -- instead of the batched kernels'
    lane-per-polynomial: the oracle's values at random and structured points, and the batched kernels' (HB_NO_EVAL_FEW=1)"""
    rnd = random.Random(p % 4093)
    for n, d, c in [(256, 86, 1), (64, 22, 1), (64, 22, 8), (100, 34, 3), (16, 8, 2), (200, 200, 1), (31, 9, 7), (130, 129, 2)]:
        if n >= p:
            n, d = p - 1, min(d, p - 1)
        x = list(range(1, n + 1))
        if rnd.random() < 0.5:
            pts = set()
            while len(pts) < n:
                pts.add(rnd.randrange(1, min(p, 1 << 70)))
            x = sorted(pts)
            rnd.shuffle(x)
        polys = rand_rows(rnd, p, c, d)
        polys[0][-1] = p - 1
        if c > 1:
            polys[1] = [0] * d
        want = oracle.vandermonde_batch_evaluate(x, polys, p)
        clear_hook(monkeypatch, "HB_NO_EVAL_FEW")
        assert hip.vandermonde_batch_evaluate(x, polys, p) == want
        if x[0] == 1 and x[-1] == n and n <= 100:            # (the batched kernels tabulate a point set first: the ones other tests have tabulated)
            set_hook(monkeypatch, "HB_NO_EVAL_FEW", "1")
            assert hip.vandermonde_batch_evaluate(x, polys, p) == want


def test_vandermonde_edge_cases(hip):
    # ragged rows are zero padded to the longest (pyx:217,232-233); tuples accepted; values reduced mod p
    x = [1, 2, 3]
    ragged = [[1], (1, 2), [5, 6, 7]]
    assert hip.vandermonde_batch_evaluate(x, ragged, P) == oracle.vandermonde_batch_evaluate(x, ragged, P)
    assert hip.vandermonde_batch_evaluate(x, [[P + 1, 2 * P + 3]], P) == [[4, 7, 10]]
    big_x = [P - 1, P - 2, 5]
    polys = [[3, 1, 4], [1, 5, 9]]
    assert hip.vandermonde_batch_evaluate(big_x, polys, P) == oracle.vandermonde_batch_evaluate(big_x, polys, P)
    with pytest.raises(OverflowError):
        hip.vandermonde_batch_evaluate(x, [[-1]], P)
    with pytest.raises(ValueError):
        hip.vandermonde_batch_evaluate(5, [[1]], P)
    with pytest.raises(hip.InterpolationError):
        hip.vandermonde_batch_interpolate([1, 1], [[1, 2]], P)
    with pytest.raises(hip.InterpolationError):
        hip.vandermonde_batch_interpolate([1, 1 + P - P, 2][:2] + [1], [[1, 2, 3]], P)
    assert hip.lagrange_interpolate([1, 2], [1, 2], P) == [0, 1]
    assert hip.lagrange_interpolate([1, 2, 3], [7, 7, 7], P) == [7]
    assert hip.lagrange_interpolate([1, 2], [0, 0], P) == []
    assert hip.evaluate([1, 2, 3, 4], 5, P) == 586
    assert hip.vandermonde_inverse([1, 2], 13) == oracle.vandermonde_inverse([1, 2], 13)


def test_golden_vandermonde_hip(hip, golden):
    g = golden("vandermonde.json")
    for case in g["cases"]:
        p, x = case["p"], case["x"]
        assert hip.vandermonde_batch_evaluate(x, case["polys"], p) == case["evals"]
        xz = [x[z] for z in case["z"]]
        ys = [[row[z] for z in case["z"]] for row in case["evals"]]
        assert hip.vandermonde_batch_interpolate(xz, ys, p) == case["interp"]
    ev = g["evaluate"]
    assert [hip.evaluate(ev["coeffs"], xv, ev["p"]) for xv in ev["xs"]] == ev["ys"]


# ------------------------------------------------------------------ strided views + in-kernel validation
def test_matvec_views_and_check():
    import torch

    from honeybadgermpc_amd._capi import Context, HbView, np_ptr

    rnd = random.Random(5)
    ctx = Context.get(P)
    lib = ctx.lib
    n, d, c = 10, 4, 333
    x = list(range(1, n + 1))
    polys = rand_rows(rnd, P, c, d)
    want = oracle.vandermonde_batch_evaluate(x, polys, P)
    V = ctypes.c_void_p()
    ctx.check(lib.hb_vand_matrix_create(ctx.h, np_ptr(ctx.host_elems(x)), n, d, ctypes.byref(V), ctx.stream()), "V")
    din = ctx.upload_ints([v for r in polys for v in r])
    # party-major output: out[i][c]
    dout = ctx.empty(n * c)
    ctx.check(lib.hb_matvec(ctx.h, V, ctx.ptr(din), HbView(d, 1), None, ctx.ptr(dout), HbView(1, c), c, ctx.stream()), "mv")
    got = ctx.download_ints(dout)
    assert [got[i * c : (i + 1) * c] for i in range(n)] == [[want[k][i] for k in range(c)] for i in range(n)]
    # decode from a row subset of the party-major buffer (IncrementalDecoder's arrival set)
    z = [7, 2, 9, 4]
    Vi = ctypes.c_void_p()
    ctx.check(lib.hb_vand_inverse_create(ctx.h, np_ptr(ctx.host_elems([x[i] for i in z])), d, ctypes.byref(Vi), ctx.stream()), "Vi")
    dec = ctx.empty(d * c)
    za = np.array(z, dtype=np.int32)
    ctx.check(lib.hb_matvec(ctx.h, Vi, ctx.ptr(dout), HbView(1, c), np_ptr(za), ctx.ptr(dec), HbView(1, c), c, ctx.stream()), "dec")
    got = ctx.download_ints(dec)
    assert [[got[l * c + k] for l in range(d)] for k in range(c)] == polys
    # validating re-encode: no mismatch on clean data, mismatch when a checked row is corrupted,
    # and no mismatch when the corrupted row is not in the check set
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    rows = np.array([0, 1, 3, 5], dtype=np.int32)

    def run_check(buf):
        flag.zero_()
        ctx.check(lib.hb_matvec_check(ctx.h, V, ctx.ptr(dec), HbView(1, c), None, ctx.ptr(buf), HbView(1, c),
                                      np_ptr(rows), len(rows), ctx.ptr(flag), c, ctx.stream()), "chk")
        return int(flag.item())

    assert run_check(dout) == 0
    bad = dout.clone()
    bad[5 * c + 77, 0] ^= 1
    assert run_check(bad) == 1
    bad = dout.clone()
    bad[6 * c + 77, 2] ^= 4
    assert run_check(bad) == 0


# ------------------------------------------------------------------ per-party open pipeline
@pytest.mark.parametrize("matrix_cores", [True, False])
@pytest.mark.parametrize("n,t,b,use_omega", [(4, 1, 3, False), (16, 5, 100, False), (16, 5, 96, True), (64, 21, 1000, False), (7, 2, 1, False),
                                             (64, 21, 700, True), (256, 85, 300, True), (100, 33, 150, True), (4, 1, 5, True),
                                             (64, 21, 22 * 16 * 3 + 5, False), (40, 13, 333, False), (48, 31, 200, False), (128, 42, 260, False),
                                             (100, 9, 10 * 16 * 5 + 3, False), (128, 7, 999, False), (80, 15, 640, False)])
def test_batch_open_vs_oracle(n, t, b, use_omega, matrix_cores):
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint

    rnd = random.Random(n * 1000 + b)
    d = t + 1
    c = (b + d - 1) // d
    point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
    x = [point(i).value for i in range(n)]
    ctx = Context.get(P)
    shares = [rnd.randrange(P) for _ in range(b)]
    # consistent received columns: column j = evaluations at x_j of random degree-t chunk polynomials
    polys1 = rand_rows(rnd, P, c, d)
    polys2 = rand_rows(rnd, P, c, d)
    e1 = oracle.vandermonde_batch_evaluate(x, polys1, P)
    e2 = oracle.vandermonde_batch_evaluate(x, polys2, P)
    r1_cols = [[e1[k][j] for k in range(c)] for j in range(n)]
    r2_cols = [[e2[k][j] for k in range(c)] for j in range(n)]
    order = list(range(n))
    rnd.shuffle(order)
    z, zc = order[:d], order[d : d + t]
    to_limbs = lambda rows: oracle._limbs([v for r in rows for v in r], P)  # noqa: E731
    rc, o_r1, o_r2msg, o_res = oracle.batch_open_limbs(P, n, d, x, oracle._limbs(shares, P), to_limbs(r1_cols), to_limbs(r2_cols), z, zc)
    assert rc == 0
    op = BatchOpen(P, n, t, z=z, zc=zc, use_omega_powers=use_omega, max_shares=b)
    op.set_matrix_cores(matrix_cores)
    # the int8 matrix-core kernels serve the points 1..n while every power fits 16 signed base-256 digits and t < 32
    eligible = (not use_omega) and t + 1 <= 32 and n ** t < 127 * 256 ** 15 and not os.environ.get("HB_NO_MFMA")
    # everything else (omega-power points, powers beyond 2^127) runs on the full-size matrix-core kernel from 4 x 4 up
    eligible_wide = (not eligible) and n >= 4 and t + 1 >= 4 and t + 1 <= 128 and not os.environ.get("HB_NO_MFMA") and not os.environ.get("HB_NO_MFMA_WIDE")
    assert op.uses_matrix_cores() == (matrix_cores and (eligible or eligible_wide))
    r1_out = op.r1_encode(ctx.upload_ints(shares))
    r2_msg = op.r1_decode(ctx.upload_ints([v for col in r1_cols for v in col]), b)
    result = op.r2_decode(ctx.upload_ints([v for col in r2_cols for v in col]), b)
    assert op.ok()
    as_np = lambda tns: tns.cpu().numpy().view(np.uint64)  # noqa: E731
    assert np.array_equal(as_np(r1_out), o_r1)
    assert np.array_equal(as_np(r2_msg), o_r2msg)
    assert np.array_equal(as_np(result), o_res)
    flat2 = [v for row in polys2 for v in row][:b]
    assert ctx.download_ints(result) == flat2
    # a corrupted validated column is detected, a corrupted non-validated one is not
    if zc:
        bad = [list(col) for col in r2_cols]
        bad[zc[-1]][c - 1] = (bad[zc[-1]][c - 1] + 5) % P
        op.r2_decode(ctx.upload_ints([v for col in bad for v in col]), b)
        assert not op.ok()
        assert op.ok()  # flag resets
    # opt-in: re-encode only tiles holding compared rows -- identical decisions
    op.set_validate_arrived_only(True)
    res2 = op.r2_decode(ctx.upload_ints([v for col in r2_cols for v in col]), b)
    assert op.ok() and np.array_equal(as_np(res2), o_res)
    if zc:
        bad = [list(col) for col in r2_cols]
        bad[zc[0]][0] = (bad[zc[0]][0] + 1) % P
        op.r2_decode(ctx.upload_ints([v for col in bad for v in col]), b)
        assert not op.ok()
    op.set_validate_arrived_only(False)
    rest = [i for i in range(n) if i not in z and i not in zc]
    if rest:
        bad = [list(col) for col in r2_cols]
        bad[rest[0]][0] = (bad[rest[0]][0] + 1) % P
        op.r2_decode(ctx.upload_ints([v for col in bad for v in col]), b)
        assert op.ok()
    # full-size entries decode and validate in one launch (include/hbmpc_hip.h, HB_OPEN_OPT_FUSED_VALIDATE); switched off
    # the plan decodes, re-encodes all n points and compares: same values, same decisions
    fused_off = bool(os.environ.get("HB_NO_FUSED_VALIDATE") or os.environ.get("HB_NO_MFMA_WIDE") or os.environ.get("HB_NO_MFMA_DECODE"))   # A/B hooks
    assert op.uses_fused_validate() == (matrix_cores and not fused_off and (eligible_wide or (eligible and d >= 4 and n >= 4)))
    if matrix_cores and eligible and d >= 4 and not fused_off:
        op.set_fused_validate(True)     # small-entry plans build the fused matrices on request
        assert op.uses_fused_validate()
    if op.uses_fused_validate():
        op.set_fused_validate(False)
        assert not op.uses_fused_validate()
        msg3 = op.r1_decode(ctx.upload_ints([v for col in r1_cols for v in col]), b)
        res3 = op.r2_decode(ctx.upload_ints([v for col in r2_cols for v in col]), b)
        assert op.ok() and np.array_equal(as_np(msg3), o_r2msg) and np.array_equal(as_np(res3), o_res)
        for fused in (False, True):
            op.set_fused_validate(fused)
            for which in ("r1", "r2"):
                for row in (zc[0], zc[-1]) if zc else ():
                    bad = [list(col) for col in (r1_cols if which == "r1" else r2_cols)]
                    k = rnd.randrange(c)
                    bad[row][k] = (bad[row][k] ^ (1 << rnd.randrange(255))) % P
                    if bad[row][k] == (r1_cols if which == "r1" else r2_cols)[row][k]:
                        continue
                    cols = ctx.upload_ints([v for col in bad for v in col])
                    (op.r1_decode if which == "r1" else op.r2_decode)(cols, b)
                    assert not op.ok(), (fused, which, row, k)


def test_full_size_properties_cfg3():
    """BASELINE config 3 size (n=64, t=21, 2^20 shares): encode -> erase -> decode round trip on
    random arrival sets, linearity of the encoder, and validation consistency."""
    import torch

    from honeybadgermpc_amd._capi import Context, HbView, np_ptr

    ctx = Context.get(P)
    lib = ctx.lib
    n, t, b = 64, 21, 1 << 20
    d = t + 1
    c = (b + d - 1) // d
    x = list(range(1, n + 1))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)

    def rand(count):
        v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
        v[:, 3] &= (1 << 61) - 1
        return v

    a, bvec = rand(c * d), rand(c * d)
    av = a.view(c, d, 4)                 # the messages uniform draws never produce: zero, constant, short, leading coefficient only, padded last chunk
    av[0] = 0
    av[1, 1:] = 0
    av[2, d // 2:] = 0
    av[3, : d - 1] = 0
    av[c - 1, 12:] = 0
    V = ctypes.c_void_p()
    ctx.check(lib.hb_vand_matrix_create(ctx.h, np_ptr(ctx.host_elems(x)), n, d, ctypes.byref(V), ctx.stream()), "V")

    def encode(v):
        out = ctx.empty(n * c)
        ctx.check(lib.hb_matvec(ctx.h, V, ctx.ptr(v), HbView(d, 1), None, ctx.ptr(out), HbView(1, c), c, ctx.stream()), "enc")
        return out

    ea, eb = encode(a), encode(bvec)
    rnd = random.Random(3)
    for trial in range(3):
        z = rnd.sample(range(n), d)      # any d surviving parties reconstruct (n - d erased)
        Vi = ctypes.c_void_p()
        ctx.check(lib.hb_vand_inverse_create(ctx.h, np_ptr(ctx.host_elems([x[i] for i in z])), d, ctypes.byref(Vi), ctx.stream()), "Vi")
        dec = ctx.empty(c * d)
        ctx.check(lib.hb_matvec(ctx.h, Vi, ctx.ptr(ea), HbView(1, c), np_ptr(np.array(z, dtype=np.int32)), ctx.ptr(dec), HbView(d, 1), c, ctx.stream()), "dec")
        assert torch.equal(dec, a)
    # linearity: enc(a) + enc(b) == enc(a + b)  (mod p), checked on the CPU with Python ints on a sample
    #   a + b is formed on the host for 200 random chunks only (exact ints)
    idx = [rnd.randrange(c) for _ in range(200)]
    ai = ctx.download_ints(a.view(c, d, 4)[idx].reshape(-1, 4))
    bi = ctx.download_ints(bvec.view(c, d, 4)[idx].reshape(-1, 4))
    s = ctx.upload_ints([(u + v) % P for u, v in zip(ai, bi)])
    es = ctx.empty(n * len(idx))
    ctx.check(lib.hb_matvec(ctx.h, V, ctx.ptr(s), HbView(d, 1), None, ctx.ptr(es), HbView(1, len(idx)), len(idx), ctx.stream()), "enc")
    ea_s = ctx.download_ints(ea.view(n, c, 4)[:, idx].reshape(-1, 4))
    eb_s = ctx.download_ints(eb.view(n, c, 4)[:, idx].reshape(-1, 4))
    assert ctx.download_ints(es) == [(u + v) % P for u, v in zip(ea_s, eb_s)]
    # spot check against the oracle on those chunks
    want = oracle.vandermonde_batch_evaluate(x, [ai[k * d : (k + 1) * d] for k in range(len(idx))], P)
    assert [ea_s[i * len(idx) + k] for k in range(len(idx)) for i in range(n)] == [v for row in want for v in row]
    # every output is canonical (< p): top limb bound check on the whole buffer
    top = ea[:, 3].cpu().numpy().view(np.uint64)
    assert int(top.max()) <= (P >> 192)


@pytest.mark.parametrize("prime", [(1 << 255) - 19, (1 << 256) - (1 << 32) - 977, (1 << 254) + 79, (1 << 254) - 245, (1 << 61) - 1])
def test_batch_open_other_moduli(prime):
    """The matrix-core path sizes its Barrett constants for 2^254 <= p < 2^256; other moduli must fall back.
    Either way the open equals the oracle's."""
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    n, t, b = 64, 21, 22 * 40 + 7
    d = t + 1
    c = (b + d - 1) // d
    rnd = random.Random(prime % 1000003)
    x = list(range(1, n + 1))
    ctx = Context.get(prime)
    shares = [rnd.randrange(prime) for _ in range(b)]
    polys1 = rand_rows(rnd, prime, c, d)
    polys2 = rand_rows(rnd, prime, c, d)
    e1 = oracle.vandermonde_batch_evaluate(x, polys1, prime)
    e2 = oracle.vandermonde_batch_evaluate(x, polys2, prime)
    r1_cols = [[e1[k][j] for k in range(c)] for j in range(n)]
    r2_cols = [[e2[k][j] for k in range(c)] for j in range(n)]
    order = list(range(n))
    rnd.shuffle(order)
    z, zc = order[:d], order[d : d + t]
    to_limbs = lambda rows: oracle._limbs([v for r in rows for v in r], prime)  # noqa: E731
    rc, o_r1, o_r2msg, o_res = oracle.batch_open_limbs(prime, n, d, x, oracle._limbs(shares, prime), to_limbs(r1_cols), to_limbs(r2_cols), z, zc)
    assert rc == 0
    op = BatchOpen(prime, n, t, z=z, zc=zc, max_shares=b)
    assert op.uses_matrix_cores() == (prime >= 1 << 254 and not os.environ.get("HB_NO_MFMA"))
    assert ctx.n_limbs == (1 if prime < 1 << 64 else 4)      # the 61-bit prime runs the 1-limb (3-digit) instantiation
    for on in (True, False):
        op.set_matrix_cores(on)
        r1_out = op.r1_encode(ctx.upload_ints(shares))
        r2_msg = op.r1_decode(ctx.upload_ints([v for col in r1_cols for v in col]), b)
        result = op.r2_decode(ctx.upload_ints([v for col in r2_cols for v in col]), b)
        assert op.ok()
        assert ctx.download_ints(r1_out) == oracle._ints(o_r1)
        assert ctx.download_ints(r2_msg) == oracle._ints(o_r2msg)
        assert ctx.download_ints(result) == oracle._ints(o_res)
        bad = [list(col) for col in r2_cols]
        bad[zc[1]][c - 1] = (bad[zc[1]][c - 1] + 1) % prime
        op.r2_decode(ctx.upload_ints([v for col in bad for v in col]), b)
        assert not op.ok()


@pytest.mark.parametrize("n,d,chunks,kind", [(64, 22, 37, "random"), (64, 22, 16, "extreme"), (64, 22, 19, "short"), (16, 6, 50, "random"),
                                             (21, 9, 19, "random"), (40, 11, 16 * 70 + 3, "random"), (22, 22, 33, "extreme"), (1, 1, 5, "random"),
                                             (100, 10, 16 * 40 + 5, "random"), (128, 8, 33, "extreme"), (80, 16, 16 * 33 + 1, "short")])
def test_matrix_core_matvec_vs_python_ints(n, d, chunks, kind):
    """The int8 matrix-core mat-vec on its own (hb_debug_mm8_*) against exact Python integers: random inputs, inputs
    at the edges of the byte-split arithmetic (0, 1, p-1, 2^256-1, 0x80.., 0x7f..), a zero-padded tail, ragged tiles."""
    from honeybadgermpc_amd._capi import Context, np_ptr

    if os.environ.get("HB_NO_MFMA"):
        pytest.skip("matrix-core path disabled by HB_NO_MFMA")
    ctx = Context.get(P)
    lib = ctx.lib
    rnd = random.Random(n * 131 + d)
    pts = list(range(1, n + 1))
    h = ctypes.c_void_p()
    ctx.check(lib.hb_debug_mm8_create(ctx.h, np_ptr(ctx.host_elems(pts)), n, d, ctypes.byref(h)), "mm8 table")
    if kind == "extreme":
        pool = [0, 1, P - 1, (1 << 256) - 1, 1 << 255, int("80" * 32, 16), int("7f" * 32, 16), int("ff00" * 16, 16)]
        xs = [rnd.choice(pool) for _ in range(chunks * d)]
        arr = np.zeros((chunks * d, 4), dtype=np.uint64)
        for k, v in enumerate(xs):
            for j in range(4):
                arr[k, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        x_dev = ctx.to_device(arr)
    else:
        xs = [rnd.randrange(P) for _ in range(chunks * d)]
        x_dev = ctx.upload_ints(xs)
    in_count = chunks * d - (7 if kind == "short" and chunks * d > 7 else 0)
    out = ctx.empty(chunks * n)
    ctx.check(lib.hb_debug_mm8_apply(ctx.h, h, ctx.ptr(x_dev), d, 1, in_count, ctx.ptr(out), n, 1, chunks * n, chunks, None, None), "mm8 apply")
    got = ctx.download_ints(out)
    for c in range(chunks):
        for i in range(n):
            want = sum(pow(pts[i], l, P) * (xs[c * d + l] if c * d + l < in_count else 0) for l in range(d)) % P
            assert got[c * n + i] == want, (c, i)


def test_full_size_open_matrix_cores_vs_valu_cfg3():
    """BASELINE config 3 size through the open plan: the matrix-core kernels and the integer-VALU kernels
    give bit-identical encodes and reconstructions, the reconstruction returns the encoded chunks, and a
    single flipped bit in a validated column is caught by both."""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    ctx = Context.get(P)
    n, t, b = 64, 21, (1 << 20) - 3          # ragged last chunk
    d = t + 1
    c = (b + d - 1) // d
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234)
    shares = torch.randint(-(1 << 63), (1 << 63) - 1, (b, 4), dtype=torch.int64, device="cuda", generator=gen)
    shares[:, 3] &= (1 << 61) - 1            # canonical values (< p)
    rnd = random.Random(8)
    order = list(range(n))
    rnd.shuffle(order)
    z, zc = order[:d], order[d : d + t]
    op = BatchOpen(P, n, t, z=z, zc=zc, max_shares=b)
    if os.environ.get("HB_NO_MFMA"):
        pytest.skip("matrix-core path disabled by HB_NO_MFMA")
    assert op.uses_matrix_cores()
    enc_m = op.r1_encode(shares)
    op.set_matrix_cores(False)
    assert not op.uses_matrix_cores()
    enc_v = op.r1_encode(shares)
    assert torch.equal(enc_m, enc_v)
    assert int(enc_m[:, 3].cpu().numpy().view(np.uint64).max()) <= (P >> 192)     # canonical outputs
    # the encode of the shares IS a consistent set of received columns: decoding it returns the shares
    for on in (True, False):
        op.set_matrix_cores(on)
        msg = op.r1_decode(enc_m, b)
        res = op.r2_decode(enc_m, b)
        assert op.ok()
        assert torch.equal(res, shares)
        assert torch.equal(msg, shares.view(-1, 4)[0::d][:c]) if b % d == 0 else torch.equal(msg[: c - 1], shares[0 : (c - 1) * d : d])
        bad = enc_m.clone()
        bad[zc[3] * c + (c - 1), 1] ^= 1 << 17          # last (ragged) chunk of a validated column
        op.r2_decode(bad, b)
        assert not op.ok()
        bad = enc_m.clone()
        bad[zc[0] * c + 12345, 0] ^= 1
        op.r1_decode(bad, b)
        assert not op.ok()
        rest = [i for i in range(n) if i not in z and i not in zc]
        bad = enc_m.clone()
        bad[rest[0] * c + 5, 0] ^= 1                     # never-arrived column: not looked at
        op.r2_decode(bad, b)
        assert op.ok()


def test_batch_open_pipeline_two_in_flight():
    """Two plans on two streams used round-robin give the same results as one open at a time."""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen, BatchOpenPipeline

    ctx = Context.get(P)
    n, t, b = 64, 21, 22 * 500 + 9
    d = t + 1
    c = (b + d - 1) // d
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    batches = []
    for _ in range(4):
        v = torch.randint(-(1 << 63), (1 << 63) - 1, (b, 4), dtype=torch.int64, device="cuda", generator=gen)
        v[:, 3] &= (1 << 61) - 1
        batches.append(v)
    z, zc = list(range(5, 5 + d)), list(range(30, 30 + t))
    ref = BatchOpen(P, n, t, z=z, zc=zc, max_shares=b)
    want = []
    for v in batches:
        enc = ref.r1_encode(v)
        want.append((enc.clone(), ref.r1_decode(enc, b).clone(), ref.r2_decode(enc, b).clone()))
    assert ref.ok()
    torch.cuda.synchronize()
    pipe = BatchOpenPipeline(P, n, t, depth=2, z=z, zc=zc, max_shares=b)
    got = []
    for v in batches:
        lane = pipe.next()
        with lane.on_stream():
            enc = lane.op.r1_encode(v)
            got.append((enc, lane.op.r1_decode(enc, b), lane.op.r2_decode(enc, b)))
    assert pipe.ok()
    torch.cuda.synchronize()
    for (a0, a1, a2), (b0, b1, b2), v in zip(want, got, batches):
        assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(b2, v)
    # a corrupted validated column on one lane is reported
    lane = pipe.next()
    with lane.on_stream():
        enc = lane.op.r1_encode(batches[0])
        enc[zc[2] * c + 3, 0] ^= 2
        lane.op.r2_decode(enc, b)
    assert not pipe.ok()
    assert pipe.ok()


# ------------------------------------------------------------------ FFT path
def test_fft_reference_vectors(hip, golden):
    assert hip.fft([0, 1], 5, 13, 4) == [1, 5, 12, 8]           # reference tests/test_ntl.py:57-68
    c = golden("constants.json")
    omega, n, d, k = c["omega"]["32"], 32, 20, 25
    rnd = random.Random(77)
    coeffs = [rnd.randrange(P) for _ in range(d)]
    want = [sum(cj * pow(pow(omega, i, P), j, P) for j, cj in enumerate(coeffs)) % P for i in range(n)]
    assert hip.fft(coeffs, omega, P, n) == want                  # test_ntl.py:71-87
    assert hip.partial_fft(coeffs, omega, P, n, k) == want[:k]   # test_ntl.py:119-136
    batch = rand_rows(rnd, P, 64, d)
    assert hip.fft_batch_evaluate(batch, omega, P, n, k) == oracle.fft_batch_evaluate(batch, omega, P, n, k)  # :90-116
    om8 = c["omega"]["8"]
    zs = [3, 0, 5]
    polys = [[1, 2, 0], [3, 2, 1], [3, 4, 2]]
    ys = [[sum(pl[i] * pow(pow(om8, z, P), i, P) for i in range(3)) % P for z in zs] for pl in polys]
    assert hip.fft_batch_interpolate(zs, ys, om8, P, 8) == polys  # test_ntl.py:159-179
    assert hip.fft_interpolate([3, 0], ys[0][:2], om8, P, 8) == [1, 2]  # :139-156 (2x+1 through 2 points)


@pytest.mark.parametrize("n,d,k,c", [(2, 1, 2, 3), (4, 2, 4, 70), (16, 6, 16, 300), (16, 16, 11, 5), (32, 20, 25, 64),
                                     (64, 22, 64, 130), (128, 34, 100, 9), (256, 86, 256, 7), (1024, 300, 1000, 2),
                                     (4096, 4096, 4096, 1), (8192, 100, 8192, 1), (16, 40, 16, 3),
                                     (4096, 100, 300, 2), (8192, 8192, 5000, 2), (1 << 16, 1 << 16, 1 << 16, 1), (1 << 16, 70000, 777, 1),
                                     (1 << 20, 1 << 20, 1 << 20, 1), (1 << 20, 3000, 1 << 20, 1)])
def test_fft_vs_oracle(hip, golden, n, d, k, c):
    """covers the mat-vec route (small n), the LDS NTT (n <= 2048), the four-step transform over it (4096 <= n <= 2^22: the
    reference benchmarks fft up to 2^20, benchmark/test_benchmark_polynomial.py:22-48), partial outputs, few coefficients, and
    truncation of coefficient lists longer than n (rsdecode_impl.h:173)"""
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import get_omega

    omega = get_omega(GF(P), n, seed=0).value
    rnd = random.Random(n * 7 + d)
    coeffs = rand_rows(rnd, P, c, d)
    coeffs[0] = [0] * d
    assert hip.fft_batch_evaluate(coeffs, omega, P, n, k) == oracle.fft_batch_evaluate(coeffs, omega, P, n, k)


def test_fft_four_step_equals_the_stage_loop_and_narrow_contexts(hip, monkeypatch):
    """the four-step transform against the stage-by-stage loop it replaces (HB_NTT_STAGE_LOOP=1), and on a 1-limb context
    (p = 2^64 - 2^32 + 1, 2-adicity 32)"""
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import get_omega

    n = 1 << 14
    omega = get_omega(GF(P), n, seed=0).value
    rnd = random.Random(14)
    rows = rand_rows(rnd, P, 2, 9000)
    got = hip.fft_batch_evaluate(rows, omega, P, n, n)
    set_hook(monkeypatch, "HB_NTT_STAGE_LOOP", "1")
    assert hip.fft_batch_evaluate(rows, omega, P, n, n) == got
    monkeypatch.delenv("HB_NTT_STAGE_LOOP")
    gold = (1 << 64) - (1 << 32) + 1
    g = 7
    om = pow(g, (gold - 1) // 8192, gold)
    assert pow(om, 4096, gold) == gold - 1
    rows = rand_rows(rnd, gold, 3, 5000)
    assert hip.fft_batch_evaluate(rows, om, gold, 8192, 8192) == oracle.fft_batch_evaluate(rows, om, gold, 8192, 8192)


def test_fft_small_primes(hip):
    # p = 13: omega = 5 has order 4; p = 97: 2-adicity 5
    assert hip.fft([1, 2, 3, 4], 5, 13, 4) == oracle.fft([1, 2, 3, 4], 5, 13, 4)
    p = 97
    g = 5
    omega = pow(g, (p - 1) // 32, p)
    assert pow(omega, 16, p) != 1
    rnd = random.Random(9)
    rows = rand_rows(rnd, p, 40, 32)
    assert hip.fft_batch_evaluate(rows, omega, p, 32, 32) == oracle.fft_batch_evaluate(rows, omega, p, 32, 32)
    zs = rnd.sample(range(32), 12)
    ys = rand_rows(rnd, p, 9, 12)
    assert hip.fft_batch_interpolate(zs, ys, omega, p, 32) == oracle.fft_batch_interpolate(zs, ys, omega, p, 32)


def test_fft_golden(hip, golden):
    for case in golden("fft.json")["cases"]:
        p, om, n = case["p"], case["omega"], case["n"]
        assert hip.fft_batch_evaluate(case["coeffs"], om, p, n, n) == case["evals"]
    for case in golden("fft_interpolate.json")["cases"]:
        assert hip.fft_batch_interpolate(case["zs"], case["ys"], case["omega"], case["p"], case["n"]) == case["coeffs"]


def test_fft_errors(hip, golden):
    om = golden("constants.json")["omega"]["8"]
    with pytest.raises(ValueError):
        hip.fft_batch_evaluate([[1, 2], [1]], om, P, 8, 8)      # ragged (UB in the reference, pyx:295)
    with pytest.raises(ValueError):
        hip.fft_batch_interpolate([1, 1], [[1, 2]], om, P, 8)    # repeated z
    with pytest.raises(ValueError):
        hip.fft_batch_interpolate([1, 9], [[1, 2]], om, P, 8)    # z out of range


def test_fft_roundtrip_full_size_cfg2(golden):
    """BASELINE config 2 size: 65 536 polynomials, n=16, t=5: evaluate -> erase 10 of 16 -> interpolate"""
    import torch

    from honeybadgermpc_amd._capi import Context, np_ptr

    ctx = Context.get(P)
    lib = ctx.lib
    n, d, c = 16, 6, 65536
    om = ctx.host_elems([golden("constants.json")["omega"]["16"]])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    a = torch.randint(-(1 << 63), (1 << 63) - 1, (c * d, 4), dtype=torch.int64, device="cuda", generator=gen)
    a[:, 3] &= (1 << 61) - 1
    ev = ctx.empty(c * n)
    ctx.check(lib.hb_fft_batch_evaluate(ctx.h, np_ptr(om), n, ctx.ptr(a), c, d, n, ctx.ptr(ev), ctx.stream()), "ev")
    zs = [11, 2, 7, 0, 15, 4]
    sel = ev.view(c, n, 4)[:, zs].contiguous().view(c * d, 4)
    back = ctx.empty(c * d)
    ctx.check(lib.hb_fft_batch_interpolate(ctx.h, np_ptr(om), n, np_ptr(np.array(zs, dtype=np.int32)), d, ctx.ptr(sel), c, ctx.ptr(back), ctx.stream()), "in")
    assert torch.equal(back, a)
    # 1 024-polynomial subset bit-exact against the oracle (cfg 2)
    sub = ctx.download_ints(a[: 1024 * d])
    want = oracle.fft_batch_evaluate([sub[i * d : (i + 1) * d] for i in range(1024)], golden("constants.json")["omega"]["16"], P, n, n)
    got = ctx.download_ints(ev[: 1024 * n])
    assert [got[i * n : (i + 1) * n] for i in range(1024)] == want


# ------------------------------------------------------------------ robust decoders
def _corrupt(rnd, word, ne, nn, p):
    word = list(word)
    idx = rnd.sample(range(len(word)), ne + nn)
    for i in range(ne):
        new = rnd.randrange(p)
        while new == word[idx[i]]:
            new = rnd.randrange(p)
        word[idx[i]] = new
    for i in range(nn):
        word[idx[i + ne]] = None
    return word, sorted(idx[:ne])


def test_gao_reference_vectors(hip):  # reference tests/test_ntl.py:196-265
    rnd = random.Random(31)
    for int_msg in ([2, 3, 2, 8, 7, 5, 9, 5], [0] * 8):
        k, n, p = 8, 22, 53
        t = k - 1
        x = list(range(n))
        encoded = [sum(int_msg[j] * pow(x[i], j, p) for j in range(k)) % p for i in range(n)]
        cmax, emax = n - 2 * t - 1, (n - 2 * t - 1) // 2
        for ne, nn in [(0, 0), (0, cmax), (emax, 0), (emax // 2, cmax // 4)]:
            for _ in range(4):
                word, _ = _corrupt(rnd, encoded, ne, nn, p)
                got = hip.gao_interpolate(x, word, k, p)
                assert got[0] == int_msg
                assert got == oracle.gao_interpolate(x, word, k, p)     # incl. the un-normalised error locator


@pytest.mark.parametrize("p,n,k", [(P, 4, 2), (P, 16, 6), (P, 64, 22), (P, 100, 34), (53, 22, 8), ((1 << 256) - 189, 31, 11)])
def test_gao_batch_vs_oracle(hip, p, n, k):
    rnd = random.Random(n * 31 + k)
    x = list(range(1, n + 1))
    emax = (n - k) // 2
    words = []
    from structured import KINDS, coordinated_errors, structured_message

    for trial in range(60):
        # zero, constant, short, padded messages (what chunk_data's padding makes of an open's last chunk) among the uniform ones
        msg = structured_message(rnd, k, p, (KINDS + ["full"])[trial % 6] if trial < 36 else None)
        enc = oracle.vandermonde_batch_evaluate(x, [msg], p)[0]
        ne = [0, emax, emax // 2, 1 if emax else 0, emax + 1, n][(trial // 6) % 6]      # the last two are beyond the radius
        ne = min(ne, n)
        if trial >= 36 and trial % 2:
            # liars that agree with each other: their symbols lie on ANOTHER polynomial of degree < k
            words.append(coordinated_errors(rnd, enc, x, k, ne, p, lambda xs, cf: oracle.vandermonde_batch_evaluate(xs, [cf], p)[0])[0])
        else:
            words.append(_corrupt(rnd, enc, ne, 0, p)[0])
    got = hip.gao_interpolate_batch(x, words, k, p)
    want = oracle.gao_interpolate_batch(x, words, k, p)
    assert got == want
    assert any(g[0] is None for g in got) or emax == 0 or p == 53


def test_gao_omega_and_erasures(hip, golden):  # reference tests/test_ntl.py:268-314
    rnd = random.Random(33)
    omega, order, n, k = golden("constants.json")["omega"]["32"], 32, 22, 8
    z = list(range(n))
    x = [pow(omega, zi, P) for zi in z]
    msg = [2, 3, 2, 8, 7, 5, 9, 5]
    enc = [sum(msg[j] * pow(x[i], j, P) for j in range(k)) % P for i in range(n)]
    for ne, nn in [(0, 0), (0, 7), (3, 0), (1, 1), (2, 3)]:
        word, _ = _corrupt(rnd, enc, ne, nn, P)
        got = hip.gao_interpolate(x, word, k, P, z=z, omega=omega, order=order, use_omega_powers=True)
        assert got[0] == msg
        assert got == oracle.gao_interpolate(x, word, k, P, z=z, omega=omega, order=order, use_omega_powers=True)


def test_wb_golden_hip(golden):
    """every reference Welch-Berlekamp outcome (incl. beyond-radius results and both failure
    messages) through the HIP kernel, batched per (n, k, p)"""
    from honeybadgermpc_amd.device import wb_decode_batch

    cases = golden("welch_berlekamp.json")["cases"]
    groups = {}
    for case in cases:
        groups.setdefault((case["p"], case["n"], case["k"]), []).append(case)
    seen = set()
    for (p, n, k), group in groups.items():
        res = wb_decode_batch(group[0]["x"], k, [c["word"] for c in group], p)
        for case, (coeffs, status) in zip(group, res):
            if case["error"] is None:
                assert status == 0 and coeffs == case["coeffs"], (p, n, k, case["word"])
            else:
                assert coeffs is None and oracle.WB_MESSAGES[status] == case["error"]
            seen.add(case["error"])
    assert seen == {None, "No solution", "found no divisors!"}


def test_wb_low_degree_messages_beyond_the_radius_hip(golden):
    """hb_wb_decode sends complete words through Gao's kernel first.  Gao decodes a message with leading zeros past
    floor((n - k) / 2) errors; the reference's Welch-Berlekamp decoder (the fixture is its output) refuses those words, so Gao's
    acceptance counts only within the radius (k_wb_take_gao) -- 41 of these 167 words came back decoded before that."""
    from honeybadgermpc_amd.device import wb_decode_batch

    cases = golden("welch_berlekamp_low_degree.json")["cases"]
    groups = {}
    for case in cases:
        groups.setdefault((case["p"], case["n"], case["k"]), []).append(case)
    for (p, n, k), group in groups.items():
        res = wb_decode_batch(group[0]["x"], k, [c["word"] for c in group], p)
        for case, (coeffs, status) in zip(group, res):
            if case["error"] is None:
                assert status == 0 and coeffs == case["coeffs"], (p, n, k, case["word"])
            else:
                assert coeffs is None and oracle.WB_MESSAGES[status] == case["error"], (p, n, k, case["word"], coeffs, status)


def test_wb_golden_cfg4_shape_hip(golden):
    """the reference's own Welch-Berlekamp outcomes at config 4's shape (n = 100, k = 34: 33 errors, erasures + errors, a
    stripped result, one word beyond the radius) through hb_wb_decode, and the error-free-erasure cases through hb_gao_decode"""
    from honeybadgermpc_amd import ntl
    from honeybadgermpc_amd.device import wb_decode_batch

    cases = golden("welch_berlekamp_cfg4.json")["cases"]
    res = wb_decode_batch(cases[0]["x"], 34, [c["word"] for c in cases], cases[0]["p"])
    for case, (coeffs, status) in zip(cases, res):
        if case["error"] is None:
            assert status == 0 and coeffs == case["coeffs"]
        else:
            assert coeffs is None and oracle.WB_MESSAGES[status] == case["error"]
    full = [c for c in cases if c["error"] is None and not any(w is None for w in c["word"])]
    got = ntl.gao_interpolate_batch(full[0]["x"], [c["word"] for c in full], 34, full[0]["p"])
    for case, (co, err) in zip(full, got):
        assert co == case["coeffs"] + [0] * (34 - len(case["coeffs"]))
        assert len(err) - 1 == len(case["errpos"])


@pytest.mark.parametrize("p,n,k", [(13, 12, 4), (53, 22, 8), (257, 70, 20), (65537, 100, 34), ((1 << 256) - 189, 100, 34), ((1 << 64) - 59, 40, 10)])
def test_gao_lazy_residues_many_words(hip, p, n, k):
    """k_gao keeps its residues lazy (a multiple of p may sit in LDS as p itself) and tests for zero accordingly: in small fields
    remainders, cofactor coefficients and quotient digits that vanish are common, at the top of the range the lazy values pass 2^256.
    Every error count 0 .. radius + 2, erased words aside (the batch entry point takes complete words), against the oracle."""
    rnd = random.Random(p % 1000 + n)
    x = list(range(1, n + 1)) if p > n else list(range(n))
    emax = (n - k) // 2
    words = []
    for trial in range(400 if n <= 40 else 160):
        msg = [rnd.randrange(p) for _ in range(k)]
        if trial % 7 == 0:
            msg = [0] * rnd.randrange(k + 1) + msg[:0]
            msg += [0] * (k - len(msg))
        if trial % 11 == 0:
            msg = [rnd.randrange(p)] + [0] * (k - 1)                 # a constant polynomial
        enc = oracle.vandermonde_batch_evaluate(x, [msg], p)[0]
        ne = min(trial % (emax + 3), n)
        words.append(_corrupt(rnd, enc, ne, 0, p)[0])
    got = hip.gao_interpolate_batch(x, words, k, p)
    want = oracle.gao_interpolate_batch(x, words, k, p)
    assert got == want


@pytest.mark.parametrize("p,n,k", [(P, 64, 22), (P, 16, 6), (P, 4, 2), (53, 22, 8), (13, 12, 4), ((1 << 256) - 189, 31, 11), ((1 << 64) - 59, 40, 10), (257, 64, 2)])
def test_gao_two_codewords_a_wave_vs_oracle(hip, monkeypatch, p, n, k):
    """k_gao_pair (point sets of at most 64 points: two codewords a wave, their narrow rounds shared) is what large batches run; HB_GAO_PAIR=1
    selects it for any batch.  Neighbouring codewords of every kind: regular ones that step together, structured messages and small fields whose
    degree anomalies make a codeword step alone, words beyond the radius that end early, an odd batch (the last wave holds one codeword)."""
    from structured import KINDS, coordinated_errors, structured_message

    set_hook(monkeypatch, "HB_GAO_PAIR", "1")
    rnd = random.Random(n * 131 + k)
    x = list(range(1, n + 1)) if p > n else list(range(n))
    emax = (n - k) // 2
    words = []
    for trial in range(181):
        msg = structured_message(rnd, k, p, (KINDS + ["full"])[trial % 6] if trial % 3 == 0 else None)
        enc = oracle.vandermonde_batch_evaluate(x, [msg], p)[0]
        ne = min([emax, emax, 0, emax // 2, 1 if emax else 0, emax + 1, emax, n, emax + 2][(trial * 7 + trial // 9) % 9], n)
        if trial % 10 == 9 and ne:
            words.append(coordinated_errors(rnd, enc, x, k, ne, p, lambda xs, cf: oracle.vandermonde_batch_evaluate(xs, [cf], p)[0])[0])
        else:
            words.append(_corrupt(rnd, enc, ne, 0, p)[0])
    got = hip.gao_interpolate_batch(x, words, k, p)
    assert got == oracle.gao_interpolate_batch(x, words, k, p)
    set_hook(monkeypatch, "HB_GAO_PAIR", "0")
    assert hip.gao_interpolate_batch(x, words, k, p) == got


def test_gao_large_batches_pair_up_by_themselves(hip, monkeypatch):
    """the default choice at 16 384 codewords of config 3's shape is two a wave: the same values as one a wave, and the oracle's on a sample"""
    n, k = 64, 22
    rnd = random.Random(77)
    x = list(range(1, n + 1))
    base = []
    for ne in (21, 21, 20, 0, 22, 7, 21, 1):
        msg = [rnd.randrange(P) for _ in range(k)]
        base.append(_corrupt(rnd, oracle.vandermonde_batch_evaluate(x, [msg], P)[0], ne, 0, P)[0])
    words = []
    for i in range(16384 + 1):
        w = list(base[(i * 5 + i // 8) % 8])
        w[i % n] = (w[i % n] + i) % P                       # (one more altered symbol: words at the radius go past it, the others stay decodable)
        words.append(w)
    monkeypatch.delenv("HB_GAO_PAIR", raising=False)
    got = hip.gao_interpolate_batch(x, words, k, P)
    assert got[:96] == oracle.gao_interpolate_batch(x, words[:96], k, P)
    assert sum(g[0] is None for g in got) > 1000 and sum(g[0] is not None for g in got) > 1000
    set_hook(monkeypatch, "HB_GAO_PAIR", "0")
    assert hip.gao_interpolate_batch(x, words, k, P) == got


@pytest.mark.parametrize("p,n,k,reps", [(P, 100, 34, 6), (P, 64, 22, 8), (53, 22, 8, 40), (13, 10, 3, 40), (P, 7, 1, 6), (P, 256, 86, 2), (P, 256, 128, 5), (P, 262, 130, 5)])
def test_wb_batch_vs_oracle(p, n, k, reps):
    from honeybadgermpc_amd.device import wb_decode_batch

    rnd = random.Random(n + k)
    x = list(range(1, n + 1))
    t = k - 1
    words = []
    for trial in range(reps):
        msg = [rnd.randrange(p) for _ in range(k)]
        enc = oracle.vandermonde_batch_evaluate(x, [msg], p)[0]
        cmax = n - 2 * t - 1
        nn = rnd.randrange(0, cmax + 1) if trial % 3 == 0 else 0
        emax = (n - nn - t) // 2
        ne = [emax, 0, emax // 2, emax + 1, min(n - nn, 2 * emax + 1)][trial % 5]
        words.append(_corrupt(rnd, enc, min(ne, n - nn), nn, p)[0])
    assert wb_decode_batch(x, k, words, p) == oracle.wb_decode_batch(x, k, words, p)


def test_sqrt_mod(hip):  # reference tests/test_ntl.py:331-341
    rnd = random.Random(0)
    for p in (P, 53, 13, (1 << 255) - 19, 0xFFFFFFFF00000001):
        xs = [rnd.randrange(p) for _ in range(200)] + [0, 1, p - 1]
        sq = [x * x % p for x in xs]
        roots = hip.sqrt_mod_batch(sq, p)
        assert [r * r % p for r in roots] == sq
        assert hip.sqrt_mod(sq[5], p) in (xs[5] % p, (p - xs[5]) % p)
    # a non-residue is reported, not silently mis-answered
    nr = next(v for v in range(2, 100) if pow(v, (P - 1) // 2, P) == P - 1)
    with pytest.raises(ValueError):
        hip.sqrt_mod(nr, P)


def test_capi_degenerate_shapes():
    """empty batches and zero-width operands through the C ABI: no launch, no error"""
    import torch

    from honeybadgermpc_amd._capi import Context, np_ptr

    ctx = Context.get(P)
    lib = ctx.lib
    x = ctx.host_elems([1, 2, 3])
    buf = ctx.empty(8)
    s = ctx.stream()
    assert lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(x), 3, ctx.ptr(buf), 0, 2, ctx.ptr(buf), s) == 0
    assert lib.hb_vandermonde_batch_interpolate(ctx.h, np_ptr(x), 3, ctx.ptr(buf), 0, ctx.ptr(buf), s) == 0
    om = ctx.host_elems([1])
    assert lib.hb_fft_batch_evaluate(ctx.h, np_ptr(om), 1, ctx.ptr(buf), 0, 1, 1, ctx.ptr(buf), s) == 0
    t = torch
    z = t.zeros(1, dtype=t.int32, device="cuda")
    z8 = t.zeros(1, dtype=t.uint8, device="cuda")
    assert lib.hb_gao_decode(ctx.h, np_ptr(x), 3, 1, ctx.ptr(buf), 0, ctx.ptr(buf), ctx.ptr(buf), ctx.ptr(z), ctx.ptr(z8), s) == 0
    assert lib.hb_wb_decode(ctx.h, np_ptr(x), 3, 1, ctx.ptr(buf), ctx.ptr(z8), 0, ctx.ptr(buf), ctx.ptr(z), ctx.ptr(z), s) == 0
    assert lib.hb_sqrt_mod(ctx.h, ctx.ptr(buf), 0, ctx.ptr(buf), ctx.ptr(z8), s) == 0
    # d = 0 coefficients: every evaluation is zero
    out = ctx.empty(6)
    out.fill_(7)
    assert lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(x), 3, ctx.ptr(buf), 2, 0, ctx.ptr(out), s) == 0
    torch.cuda.synchronize()
    assert int(out.abs().sum().item()) == 0
    # bad arguments are reported, not crashed on
    from honeybadgermpc_amd._capi import HB_ERR_BAD_ARG

    assert lib.hb_fft_batch_evaluate(ctx.h, np_ptr(om), 3, ctx.ptr(buf), 1, 1, 1, ctx.ptr(buf), s) == HB_ERR_BAD_ARG   # order not a power of two
    assert lib.hb_wb_decode(ctx.h, np_ptr(x), 3, 5, ctx.ptr(buf), ctx.ptr(z8), 1, ctx.ptr(buf), ctx.ptr(z), ctx.ptr(z), s) == HB_ERR_BAD_ARG  # k > n


def test_interpolation_points_including_zero(hip):
    """refine_triples interpolates at x = 0..d (reference progs/triple_refinement.py:43-44)"""
    rnd = random.Random(2)
    for d in (1, 4, 11):
        x = list(range(d + 1))
        rows = rand_rows(rnd, P, 5, d + 1)
        want = oracle.vandermonde_batch_interpolate(x, rows, P)
        assert hip.vandermonde_batch_interpolate(x, rows, P) == want
        more = list(range(d + 1, 2 * d + 1))
        assert hip.vandermonde_batch_evaluate(more, want, P) == oracle.vandermonde_batch_evaluate(more, want, P)


def test_gao_cofactor_pinned_by_the_references_polynomial_class(hip, golden):
    """tests/golden/gao_cofactor.json: coefficients, the un-normalised EEA cofactor and the (None, None) decisions from the reference's
    own Polynomial class running partial_gcd's recurrence (oracle/gen_golden.py section H) -- the HIP decoder (fraction-free Euclid
    + one inversion, hb_gao.hip) reproduces them coefficient for coefficient, one word at a time and as batches"""
    cases = golden("gao_cofactor.json")["cases"]
    for c in cases:
        got = hip.gao_interpolate(c["x"], c["y"], c["k"], P)
        assert got == ((c["coeffs"], c["v"]) if c["coeffs"] is not None else (None, None)), (c["kind"], c["k"], len(c["x"]))
    # words over the same points and without erasures, as one launch
    groups = {}
    for c in cases:
        if None not in c["y"]:
            groups.setdefault((tuple(c["x"]), c["k"]), []).append(c)
    assert groups
    for (x, k), cs in groups.items():
        got = hip.gao_interpolate_batch(list(x), [c["y"] for c in cs], k, P)
        for g, c in zip(got, cs):
            assert tuple(g) == ((c["coeffs"], c["v"]) if c["coeffs"] is not None else (None, None)), (c["kind"], k)


@pytest.mark.parametrize("p,n,k,words", [(P, 100, 34, 6), (P, 64, 22, 8), (P, 25, 4, 40), (53, 22, 8, 60), (13, 12, 3, 40), ((1 << 64) - 59, 40, 10, 24)])
def test_wb_batches_with_one_erasure_pattern_vs_oracle(p, n, k, words):
    """A batch whose codewords all lost the SAME symbols (the protocol's case: the parties that have not arrived, reed_solomon.py:201-204) is
    decoded by Gao's kernels over the points that are left; whatever they do not settle -- beyond the radius of the REDUCED word, too few
    points, low-degree messages -- keeps the row reduction's outcomes.  Against the oracle's restatement of the reference's decoder, with the
    structured messages of tests/structured.py, errors up to and beyond the reduced radius, coordinated liars; and against a batch with
    per-codeword patterns (the row reduction throughout)."""
    from structured import coordinated_errors, structured_message

    from honeybadgermpc_amd.device import wb_decode_batch

    rnd = random.Random(n * 131 + k)
    x = list(range(1, n + 1))
    cmax = n - 2 * (k - 1) - 1                       # erasures the reference's decoder admits (reed_solomon_wb.py:131)
    for trial in range(4):
        c = [1, max(1, cmax // 2), cmax, min(n - k, cmax + 2)][trial] if cmax > 0 else 0
        if c <= 0:
            continue
        erased = rnd.sample(range(n), c)
        emax = max(0, (n - c - k) // 2)
        rows = []
        for w in range(words):
            msg = structured_message(rnd, k, p)
            enc = oracle.vandermonde_batch_evaluate(x, [msg], p)[0]
            ne = min(n - c, [0, emax, emax // 2, emax + 1, emax + 2, 1][w % 6])
            alive = [i for i in range(n) if i not in erased]
            if w % 4 == 3:
                other = coordinated_errors(rnd, enc, x, k, 0, p, lambda xs, cf: oracle.vandermonde_batch_evaluate(xs, [cf], p)[0])[0]
                liar_vals = oracle.vandermonde_batch_evaluate(x, [[rnd.randrange(p) for _ in range(k)]], p)[0]
                for i in rnd.sample(alive, ne):
                    other[i] = liar_vals[i]
                enc = other
            else:
                for i in rnd.sample(alive, ne):
                    enc[i] = (enc[i] + rnd.randrange(1, p)) % p
            rows.append([None if i in erased else enc[i] for i in range(n)])
        got = wb_decode_batch(x, k, rows, p)
        want = oracle.wb_decode_batch(x, k, rows, p)
        assert got == want, (trial, c, next(i for i in range(words) if got[i] != want[i]))
        # the same words with ONE codeword losing another symbol: no shared pattern, the row reduction decides everything -- same outcomes
        if n - c - 1 >= k:
            mixed = [list(r) for r in rows]
            extra = next(i for i in range(n) if mixed[0][i] is not None)
            mixed[0][extra] = None
            assert wb_decode_batch(x, k, mixed, p)[1:] == want[1:]


@pytest.mark.parametrize("p,n,k,words", [(BLS, 25, 8, 400), (BLS, 100, 34, 330), ((1 << 61) - 1, 30, 7, 300)])
def test_wb_batches_with_a_few_erasure_patterns_vs_oracle(p, n, k, words):
    """Per-codeword erasure patterns (reed_solomon_wb.py:129-151 takes any): a batch with a few distinct patterns is cut by pattern; groups of at
    least 64 codewords run Gao's kernels on their reduced point sets, the rest (small groups, patterns that leave too few points, words beyond the
    reduced radius) the row reduction -- every outcome (coefficients, lengths, the reference's refusals) as the oracle's restatement of the
    reference's decoder gives it.  Patterns here: none erased, three large groups, one group of 5 codewords, one pattern with too few points."""
    from structured import structured_message

    from honeybadgermpc_amd.device import wb_decode_batch

    rnd = random.Random(n * 977 + k)
    x = list(range(1, n + 1))
    cmax = n - 2 * (k - 1) - 1
    pats = [[], rnd.sample(range(n), max(1, cmax // 3)), rnd.sample(range(n), max(1, cmax // 2)), rnd.sample(range(n), max(1, cmax)),
            rnd.sample(range(n), 2), rnd.sample(range(n), min(n - k, cmax + 3))]
    sizes = [words // 4, words // 4, words // 4, words - 3 * (words // 4) - 5 - 70, 5, 70]
    which = [gi for gi, sz in enumerate(sizes) for _ in range(sz)]
    rnd.shuffle(which)
    rows = []
    for w, gi in enumerate(which):
        erased = set(pats[gi])
        msg = structured_message(rnd, k, p)
        enc = oracle.vandermonde_batch_evaluate(x, [msg], p)[0]
        alive = [i for i in range(n) if i not in erased]
        emax = max(0, (len(alive) - k) // 2)
        ne = min(len(alive), [0, emax, emax // 2, emax + 1, 1, emax][w % 6])
        for i in rnd.sample(alive, ne):
            enc[i] = (enc[i] + rnd.randrange(1, p)) % p
        rows.append([None if i in erased else enc[i] for i in range(n)])
    got = wb_decode_batch(x, k, rows, p)
    want = oracle.wb_decode_batch(x, k, rows, p)
    bad = [i for i in range(len(rows)) if got[i] != want[i]]
    assert not bad, (len(bad), bad[:5], [which[i] for i in bad[:5]])
