"""Decode + validate at small-integer points on the small-entry matrix-core kernel (hb_mfma_fused.hip) through the C ABI:
hb_quick_dec_* (the optimistic step of IncrementalDecoder in two halves, reference reed_solomon.py:305-330), the open plans that
use the same images, and the decoder API on top -- bit-exact against the oracle's interpolate / evaluate
(reference hbmpc_ntl_helpers.pyx:139-244)."""
import ctypes
import random

import numpy as np
import pytest

import oracle
from conftest import BLS as P

pytestmark = pytest.mark.gpu

INT_MAX = (1 << 31) - 1
SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


class _QDec:
    def __init__(self, ctx, x):
        from honeybadgermpc_amd._capi import np_ptr

        self.ctx, self.n = ctx, len(x)
        self.h = ctypes.c_void_p()
        self.rc = ctx.lib.hb_quick_dec_create(ctx.h, np_ptr(ctx.host_elems(x)), len(x), ctypes.byref(self.h), ctx.stream())

    def arrivals(self, z, nc, n_coef):
        from honeybadgermpc_amd._capi import np_ptr

        za = np.array(z, dtype=np.int32)
        return self.ctx.lib.hb_quick_dec_arrivals(self.h, np_ptr(za), len(z), nc, n_coef, self.ctx.stream())

    def decide(self, zc, cols, c, n_out_elems, lo=0, hi=None):
        from honeybadgermpc_amd._capi import np_ptr

        zca = np.array(zc if zc else [0], dtype=np.int32)
        out = self.ctx.empty(n_out_elems)
        out.zero_()
        flag, first = ctypes.c_int32(-1), ctypes.c_int32(-1)
        rc = self.ctx.lib.hb_quick_dec_decide(self.h, np_ptr(zca), len(zc), self.ctx.ptr(cols), c, lo, c if hi is None else hi, self.ctx.ptr(out),
                                              ctypes.byref(flag), ctypes.byref(first), self.ctx.stream())
        self.ctx.check(rc, "hb_quick_dec_decide")
        return out, flag.value, first.value

    def close(self):
        if self.h:
            self.ctx.lib.hb_quick_dec_destroy(self.h)


def _codewords(rnd, x, d, c, p):
    from structured import structured_rows

    polys = structured_rows(rnd, p, c, d)                   # zero, constant, short, padded polynomials among the uniform ones
    enc = oracle.vandermonde_batch_evaluate(x, polys, p)                       # [c][n]
    return polys, [enc[k][j] for j in range(len(x)) for k in range(c)]


def _omega_points(n):
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint

    point = EvalPoint(GF(P), n, use_omega_powers=True)
    return [point(i).value for i in range(n)]


@pytest.mark.parametrize("p,x,t,c", [
    (P, "omega64", 21, 600),                              # omega points: full-size entries -- the same two halves on the full-size kernel
    (P, "omega256", 85, 260),
    (P, list(range(1, 101)), 33, 300),                    # 100^33 does not fit 16 digits: the full-size kernel too
    (P, list(range(1, 65)), 21, 1500),                    # config 3's shape: 43 rows = three row tiles, 12 passes for 8 waves
    (P, list(range(1, 17)), 5, 333),                      # one row tile: four passes, four waves only scale
    (P, list(range(1, 8)), 3, 70),                        # d = 4: the smallest shape
    (P, list(range(1, 41)), 12, 200),                     # d = 13: two K-blocks
    (P, [3, 50, 7, 19, 200, 101, 64, 1, 999, 12, 77, 5, 31, 444, 2, 650], 4, 129),     # distinct small integers in no order
    (P, list(range(1, 201)), 4, 150),                     # more than 128 parties: no candidate store, the compared rows are built at the end
    (SECP_N, list(range(1, 33)), 10, 900),                # p > 2^255: the scaled element is made canonical before it is packed
    ((1 << 255) + 95, list(range(1, 25)), 5, 257),
])
def test_quick_dec_vs_oracle(p, x, t, c):
    from honeybadgermpc_amd._capi import HB_OK, Context

    if pow(2, p - 1, p) != 1:
        pytest.skip("not a prime")
    if isinstance(x, str):
        x = _omega_points(int(x[5:]))
    ctx = Context.get(p)
    rnd = random.Random(len(x) * 1000 + t)
    n, d = len(x), t + 1
    polys, flat = _codewords(rnd, x, d, c, p)
    cols = ctx.upload_ints(flat)
    qd = _QDec(ctx, x)
    assert qd.rc == HB_OK
    try:
        for trial in range(4):
            order = list(range(n))
            rnd.shuffle(order)
            z = order[:d]
            zc = order[d : d + rnd.choice([0, 1, t, min(n - d, t + 2)])]
            # every coefficient, chunk-major
            assert qd.arrivals(z, len(zc), d) == HB_OK
            out, flag, first = qd.decide(zc, cols, c, c * d)
            assert ctx.download_ints(out) == [v for row in polys for v in row], (trial, z, zc)
            assert flag == 0 and first == INT_MAX
            # the constant terms only (what R1 forwards)
            assert qd.arrivals(z, len(zc), 1) == HB_OK
            out, flag, first = qd.decide(zc, cols, c, c)
            assert ctx.download_ints(out) == [row[0] for row in polys]
            assert flag == 0 and first == INT_MAX
            if not zc:
                continue
            # corrupted compared columns: the flag and the FIRST disagreeing chunk; the decode itself only reads the rows z
            bad_chunks = sorted(rnd.sample(range(c), 3))
            bad = list(flat)
            for m in bad_chunks:
                j = rnd.choice(zc)
                bad[j * c + m] = (bad[j * c + m] + 1 + rnd.randrange(p - 1)) % p
            assert qd.arrivals(z, len(zc), d) == HB_OK
            out, flag, first = qd.decide(zc, ctx.upload_ints(bad), c, c * d)
            assert flag == 1 and first == bad_chunks[0], (first, bad_chunks)
            assert ctx.download_ints(out) == [v for row in polys for v in row]
            # the verdict words are reset for the next launch of the same object
            assert qd.arrivals(z, len(zc), d) == HB_OK
            _, flag, first = qd.decide(zc, cols, c, c * d)
            assert flag == 0 and first == INT_MAX
            # a corrupted DECODED column changes that chunk's coefficients and disagrees with every compared row
            bad = list(flat)
            m = rnd.randrange(c)
            bad[z[0] * c + m] = (bad[z[0] * c + m] + 5) % p
            assert qd.arrivals(z, len(zc), 1) == HB_OK
            _, flag, first = qd.decide(zc, ctx.upload_ints(bad), c, c)
            assert flag == 1 and first == m
            if c > 100:
                # chunks [lo, hi): nothing outside is read, written or compared; the first disagreement counts from lo
                lo, hi = c // 3, c // 3 + 70
                bad = list(flat)
                bad[zc[0] * c + lo - 1] = (bad[zc[0] * c + lo - 1] + 1) % p
                bad[zc[-1] * c + lo + 9] = (bad[zc[-1] * c + lo + 9] + 1) % p
                bad[zc[-1] * c + hi] = (bad[zc[-1] * c + hi] + 1) % p
                assert qd.arrivals(z, len(zc), d) == HB_OK
                out, flag, first = qd.decide(zc, ctx.upload_ints(bad), c, c * d, lo=lo, hi=hi)
                assert flag == 1 and first == 9
                got = ctx.download_ints(out)
                assert got[lo * d : hi * d] == [v for row in polys[lo:hi] for v in row]
                assert not any(got[: lo * d]) and not any(got[hi * d :])
    finally:
        qd.close()


def test_quick_dec_says_what_it_does_not_take():
    from honeybadgermpc_amd._capi import HB_ERR_BAD_ARG, HB_ERR_UNSUPPORTED, HB_OK, Context, np_ptr

    ctx = Context.get(P)
    qd = _QDec(ctx, list(range(1, 101)))
    assert qd.rc == HB_OK
    assert qd.arrivals(list(range(34)), 33, 34) == HB_OK                       # (the full-size kernel: 100^33 does not fit 16 digits)
    assert qd.arrivals([0, 1, 2], 2, 3) == HB_ERR_UNSUPPORTED                  # fewer than four coefficients: neither kernel
    assert qd.arrivals([0, 1, 2, 2], 3, 4) == HB_ERR_BAD_ARG                   # a repeated arrival
    assert qd.arrivals([0, 1, 2, 3], 2, 4) == HB_OK
    import torch

    cols = ctx.upload_ints([1] * 100 * 8)
    zc = np.array([2, 9], dtype=np.int32)                                       # overlaps the arrivals
    flag, first = ctypes.c_int32(0), ctypes.c_int32(0)
    out = ctx.empty(8 * 4)
    rc = ctx.lib.hb_quick_dec_decide(qd.h, np_ptr(zc), 2, ctx.ptr(cols), 8, 0, 8, ctx.ptr(out), ctypes.byref(flag), ctypes.byref(first), ctx.stream())
    assert rc == HB_ERR_BAD_ARG
    zc = np.array([7, 9, 11], dtype=np.int32)                                   # not the number of compared senders announced
    assert qd.arrivals([0, 1, 2, 3], 2, 4) == HB_OK
    rc = ctx.lib.hb_quick_dec_decide(qd.h, np_ptr(zc), 3, ctx.ptr(cols), 8, 0, 8, ctx.ptr(out), ctypes.byref(flag), ctypes.byref(first), ctx.stream())
    assert rc == HB_ERR_BAD_ARG
    torch.cuda.synchronize()
    qd.close()
    assert _QDec(ctx, [1, 2, 3, 3, 5, 6, 7, 8]).rc == HB_ERR_UNSUPPORTED        # repeated points
    narrow = Context.get((1 << 61) - 1)
    assert _QDec(narrow, list(range(1, 9))).rc == HB_ERR_UNSUPPORTED


@pytest.mark.parametrize("n,t", [(64, 21), (16, 5), (31, 10)])
def test_open_plan_small_and_wide_kernels_agree(n, t):
    """a plan at the production points decodes + validates on the small-entry kernel by default; the full-size kernel on request
    and the unfused pipeline give the same words and the same verdicts"""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    ctx = Context.get(P)
    rnd = random.Random(n + t)
    d = t + 1
    c = 700
    x = list(range(1, n + 1))
    polys, flat = _codewords(rnd, x, d, c, P)
    cols = ctx.upload_ints(flat)
    order = list(range(n))
    rnd.shuffle(order)
    z, zc = order[:d], order[d : d + t]
    op = BatchOpen(P, n, t, z=z, zc=zc, max_shares=c * d)
    assert op.fused_validate_kernel() == "small"
    want = [v for row in polys for v in row]
    outs = {}
    for kernel in (True, "wide", False):
        op.set_fused_validate(kernel)
        assert op.fused_validate_kernel() == {True: "small", "wide": "wide", False: None}[kernel]
        res = op.r2_decode(cols, c * d)
        msg = op.r1_decode(cols, c * d)
        assert op.ok()
        assert ctx.download_ints(res) == want and ctx.download_ints(msg) == [row[0] for row in polys], kernel
        outs[kernel] = (res, msg)
        bad = cols.clone().view(n, c, 4)
        bad[zc[-1], c - 1, 0] += 1
        op.r2_decode(bad.view(n * c, 4), c * d)
        assert not op.ok(), kernel
        op.r1_decode(bad.view(n * c, 4), c * d)
        assert not op.ok(), kernel
    assert torch.equal(outs[True][0], outs["wide"][0]) and torch.equal(outs[True][1], outs[False][1])


def test_decoder_receives_in_place_and_forwards_constant_terms():
    """DeviceIncrementalDecoder(columns=...): the transport's party-major buffer is decoded where it lies; want="constant" yields
    the R2 message alone; both agree with the host decoder's results"""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder

    ctx = Context.get(P)
    rnd = random.Random(77)
    n, t, c = 64, 21, 900
    d = t + 1
    x = list(range(1, n + 1))
    polys, flat = _codewords(rnd, x, d, c, P)
    cols = ctx.upload_ints(flat).view(n, c, 4)
    for want in ("all", "constant"):
        order = list(range(n))
        rnd.shuffle(order)
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=c, columns=cols, want=want)
        used = 0
        for idx in order:
            assert dec.slot(idx).data_ptr() == cols[idx].data_ptr()
            dec.add(idx)
            used += 1
            if dec.done():
                break
        assert used == d + t and dec.quick_launches == 1
        res, errs = dec.get_results()
        assert errs == set()
        if want == "all":
            assert ctx.download_ints(res.reshape(-1, 4)) == [v for row in polys for v in row]
        else:
            assert tuple(res.shape) == (c, 1, 4) and ctx.download_ints(res.reshape(-1, 4)) == [row[0] for row in polys]
    # a liar among the compared senders: the constant-terms decoder falls to the robust phase and still yields every constant term
    bad = cols.clone()
    order = list(range(n))
    rnd.shuffle(order)
    liar = order[d + 3]
    bad[liar, 5, 1] ^= 1
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=c, columns=bad, want="constant")
    for idx in order:
        dec.add(idx)
        if dec.done():
            break
    res, errs = dec.get_results()
    assert errs == {liar}
    assert ctx.download_ints(res[:, 0, :].contiguous()) == [row[0] for row in polys]
    torch.cuda.synchronize()


def test_pooled_probe_starts_from_a_reset():
    """a probe handed back to the pool keeps its last owner's points; a second decoder that stalls on the same
    polynomial index with an arrival list extending the old one must not judge the first decoder's data.  Two decoders back to
    back through the probe path, same arrival order, different data, each against the host mirror."""
    import torch

    from honeybadgermpc_amd import device
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder

    ctx = Context.get(P)
    rnd = random.Random(5)
    n, t, c = 16, 5, 12
    d = t + 1
    x = list(range(1, n + 1))
    device._probe_pool.idle.clear()
    liars = [0, 1, 2, 3, 4]
    # the liars arrive spread out, one among every six consecutive arrivals: a candidate interpolated from either end of the arrival
    # list is contaminated, so no candidate stands (device.py _candidate_cap) and the probe has to decide
    honest = [i for i in range(n) if i not in liars]
    order = [0] + honest[:4] + [1] + honest[4:7] + [2, 3] + honest[7:10] + [4] + honest[10:]
    results = []
    for rep in range(2):
        polys, flat = _codewords(rnd, x, d, c, P)
        cols = ctx.upload_ints(flat).view(n, c, 4).clone()
        for j in liars:
            cols[j] = ctx.upload_ints([rnd.randrange(P) for _ in range(c)])
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=c)
        trace = []
        for idx in order:
            dec.add(idx, cols[idx])
            trace.append((dec.done(), sorted(dec._confirmed_errors)))
            if dec.done():
                break
        res, errs = dec.get_results()
        assert errs == set(liars)
        assert ctx.download_ints(res.reshape(-1, 4)) == [v for row in polys for v in row], rep
        assert dec.probes > 0
        results.append(trace)
        del dec                                                       # hands its probe back to the pool
    assert results[0] == results[1]
    assert sum(len(v) for v in device._probe_pool.idle.values()) >= 1
    torch.cuda.synchronize()
