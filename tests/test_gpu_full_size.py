"""
GPU tests at BASELINE.json's full sizes for the configurations round 1 only covered at reduced size:
config 4 (robust decoders, n=100 t=33), config 5's shard (n=256 t=85, 2^19 shares), a bounded randomised differential run
of the two kernel families, every entry point on the narrow (p < 2^64, 1-limb) instantiation, and the bounded table caches.
Bit-exact: generating polynomials / secrets must come back, and oracle subsets are compared value by value.
"""
import ctypes
import os
import random
import time

import numpy as np
import pytest

import oracle
from conftest import BLS, clear_hook, set_hook

pytestmark = pytest.mark.gpu

P = BLS


def _rand(torch, count, gen):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
    t[:, 3] &= (1 << 61) - 1
    return t


def _structure(coef, c, d):
    """in place on a (c * d, 4) tensor of uniformly random coefficients: the messages uniform draws never produce (tests/structured.py) --
    chunk 0 zero, chunk 1 constant, chunk 2 short, chunk 3 only its leading coefficient, one in the middle short, the LAST chunk padded
    with zeros as chunk_data pads the last chunk of every open (reference utils/misc.py:33-51)"""
    v = coef.view(c, d, 4)
    if c < 8 or d < 4:
        return coef
    v[0] = 0
    v[1, 1:] = 0
    v[2, d // 2:] = 0
    v[3, : d - 1] = 0
    v[c // 2, 2:] = 0
    v[c - 1, 3:] = 0
    return coef


# ---------------------------------------------------------------------------------------------- config 4
@pytest.mark.parametrize("decoder,count", [("gao", 1 << 18), ("wb", 1 << 14), ("wb", 1 << 18)])
def test_robust_decoders_full_batch_cfg4(decoder, count):
    """n=100, t=33: `count` codewords of random degree-33 polynomials, exactly 33 positions of each replaced by random field
    elements (the decoding radius), no erasures -> every generating polynomial recovered bit for bit; a 64-codeword subset
    against the oracle's decoders; a batch with codewords beyond the radius keeps the reference's outcomes."""
    import torch

    from honeybadgermpc_amd._capi import Context, np_ptr

    n, t = 100, 33
    k = t + 1
    ctx = Context.get(P)
    lib = ctx.lib
    xh = ctx.host_elems(list(range(1, n + 1)))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(44)
    msg = _structure(_rand(torch, count * k, gen), count, k)
    code = ctx.empty(count * n)
    ctx.check(lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(xh), n, ctx.ptr(msg), count, k, ctx.ptr(code), ctx.stream()), "enc")
    pos = torch.rand((count, n), device="cuda", generator=gen).argsort(dim=1)[:, :t]
    idx = (torch.arange(count, device="cuda").unsqueeze(1) * n + pos).reshape(-1)
    bad = code.clone()
    bad[idx] = _rand(torch, count * t, gen)
    out = ctx.empty(count * k)
    if decoder == "gao":
        err = ctx.empty(count * (n + 1))
        elen = torch.zeros(count, dtype=torch.int32, device="cuda")
        ok = torch.zeros(count, dtype=torch.uint8, device="cuda")
        ctx.check(lib.hb_gao_decode(ctx.h, np_ptr(xh), n, k, ctx.ptr(bad), count, ctx.ptr(out), ctx.ptr(err), ctx.ptr(elen), ctx.ptr(ok), ctx.stream()), "gao")
        assert bool(ok.all().item()) and bool((elen == t + 1).all().item())
        # the error locator's roots are exactly the corrupted positions (first 16 codewords, exact integers)
        ev = ctx.empty(16 * n)
        ctx.check(lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(xh), n, ctx.ptr(err), 16, n + 1, ctx.ptr(ev), ctx.stream()), "ev")
        roots = (ev.view(16, n, 4) == 0).all(dim=2)
        want = torch.zeros((16, n), dtype=torch.bool, device="cuda")
        want.scatter_(1, pos[:16], True)
        changed = (bad.view(count, n, 4)[:16] != code.view(count, n, 4)[:16]).any(dim=2)
        assert torch.equal(roots, want & changed)
    else:
        present = torch.ones(count * n, dtype=torch.uint8, device="cuda")
        olen = torch.zeros(count, dtype=torch.int32, device="cuda")
        st = torch.zeros(count, dtype=torch.int32, device="cuda")
        ctx.check(lib.hb_wb_decode(ctx.h, np_ptr(xh), n, k, ctx.ptr(bad), ctx.ptr(present), count, ctx.ptr(out), ctx.ptr(olen), ctx.ptr(st), ctx.stream()), "wb")
        assert bool((st == 0).all().item())
        top_nonzero = (msg.view(count, k, 4)[:, k - 1] != 0).any(dim=1)
        assert bool((olen[top_nonzero] == k).all().item())
    assert torch.equal(out, msg), "decoded coefficients differ from the generating polynomials"
    # oracle subset
    sub = 8 if decoder == "wb" else 64
    words = ctx.download_ints(bad[: sub * n])
    rows = [words[i * n : (i + 1) * n] for i in range(sub)]
    got = ctx.download_ints(out[: sub * k])
    if decoder == "gao":
        ref = oracle.gao_interpolate_batch(list(range(1, n + 1)), rows, k, P)
        assert [got[i * k : (i + 1) * k] for i in range(sub)] == [r[0] for r in ref]
    else:
        ref = oracle.wb_decode_batch(list(range(1, n + 1)), k, rows, P)
        for i in range(sub):
            coeffs = ref[i][0]
            assert coeffs is not None and got[i * k : i * k + len(coeffs)] == coeffs
    if decoder == "wb":
        # beyond the radius (34 errors): the row reduction decides, exactly as the oracle's restatement of the reference does
        m = 6
        gen2 = random.Random(6)
        words = ctx.download_ints(code[: m * n])
        rows = []
        for i in range(m):
            row = words[i * n : (i + 1) * n]
            for j in gen2.sample(range(n), t + 1 + (i % 2)):
                row[j] = gen2.randrange(P)
            rows.append(row)
        from honeybadgermpc_amd.device import wb_decode_batch

        assert wb_decode_batch(list(range(1, n + 1)), k, rows, P) == oracle.wb_decode_batch(list(range(1, n + 1)), k, rows, P)


# ---------------------------------------------------------------------------------------------- config 5 (one shard)
def test_full_size_open_cfg5_shard():
    """One GPU's 1/8 shard of BASELINE config 5 (n=256, t=85, omega points, 2^19 shares): consistent inputs, both kernel
    families agree, the secrets come back, a corrupted validated column is caught, and the first 1 024 chunks of the R1
    encode and of the result equal the oracle's."""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    import bench

    n, t, b = 256, 85, (1 << 22) // 8
    d = t + 1
    c = (b + d - 1) // d
    ctx = Context.get(P)
    shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs_light(torch, ctx, n, t, b, True, seed=55)
    order = np.random.Generator(np.random.PCG64(5)).permutation(n).tolist()
    z, zc = order[:d], order[d : d + t]
    op = BatchOpen(P, n, t, z=z, zc=zc, use_omega_powers=True, max_shares=b)
    assert op.uses_matrix_cores() or os.environ.get("HB_NO_MFMA") or os.environ.get("HB_NO_MFMA_WIDE")   # A/B hooks
    outs = {}
    for on in (True, False):
        op.set_matrix_cores(on)
        r1 = op.r1_encode(shares0)
        msg = op.r1_decode(r1_cols, b)
        res = op.r2_decode(r2_cols, b)
        assert op.ok()
        assert torch.equal(res, secrets) and torch.equal(msg, r2_cols[:c])
        outs[on] = (r1.clone(), msg.clone(), res.clone())
    assert all(torch.equal(a, bb) for a, bb in zip(outs[True], outs[False])), "matrix-core and integer-VALU decodes differ"
    bad = r2_cols.clone()
    bad[zc[3] * c + c // 2, 1] ^= 1 << 17
    op.set_matrix_cores(True)
    op.r2_decode(bad, b)
    assert not op.ok()
    # oracle subset: the first 1 024 chunks
    sub = 1024
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import EvalPoint

    point = EvalPoint(GF(P), n, use_omega_powers=True)
    sh = ctx.download_ints(shares0[: sub * d])
    want = oracle.fft_batch_evaluate([sh[i * d : (i + 1) * d] for i in range(sub)], point.omega.value, P, point.order, n)
    got = ctx.download_ints(outs[True][0].view(n, c, 4)[:, :sub].contiguous().view(n * sub, 4))
    assert [[got[j * sub + i] for j in range(n)] for i in range(sub)] == want
    cols = ctx.download_ints(r2_cols.view(n, c, 4)[:, :sub].contiguous().view(n * sub, 4))
    ys = [[cols[j * sub + i] for j in z] for i in range(sub)]
    coef = oracle.fft_batch_interpolate(z, ys, point.omega.value, P, point.order)
    assert ctx.download_ints(outs[True][2][: sub * d]) == [v for row in coef for v in row]


# ---------------------------------------------------------------------------------------------- bounded stress
@pytest.mark.parametrize("n, t, b, use_omega, spread", [(64, 21, 1 << 20, False, False), (256, 85, (1 << 22) // 8, True, False), (64, 21, 1 << 18, False, True)])
def test_full_size_decoder_under_attack(n, t, b, use_omega, spread):
    """The device decoder at config 3's / config 5's shard shape with t liars that send garbage in every chunk: arriving FIRST (the newest
    columns give a candidate that decides, device.py _candidate_cap: no incremental decode) or spread over the arrival list (no candidate
    stands: the probe decides).  Size-independent checks: every shared polynomial comes back bit for bit, exactly the liars are named, and
    nothing is final before the last needed column (with t liars among the arrivals the reference needs all n)."""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder

    d = t + 1
    c = (b + d - 1) // d
    ctx = Context.get(P)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(n + t)

    def rand(count):
        v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
        v[:, 3] &= (1 << 61) - 1
        return v

    coef = _structure(rand(c * d), c, d)
    enc = BatchOpen(P, n, t, use_omega_powers=use_omega, max_shares=c * d)
    cols = enc.r1_encode(coef).view(n, c, 4).clone()
    rng = np.random.Generator(np.random.PCG64(n))
    liars = sorted(rng.choice(n, size=t, replace=False).tolist())
    for j in liars:
        cols[j] = rand(c)
    honest = [j for j in rng.permutation(n).tolist() if j not in liars]
    if spread:
        step = len(honest) // (t + 1)
        order = []
        for i, j in enumerate(liars):
            order += honest[i * step:(i + 1) * step] + [j]
        order += honest[t * step:]
    else:
        order = liars + honest
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=c, use_omega_powers=use_omega, columns=cols)
    used = 0
    for j in order:
        assert not dec.done()
        dec.add(j)
        used += 1
        if dec.done():
            break
    res, errs = dec.get_results()
    assert used == n and errs == set(liars)
    assert torch.equal(res.reshape(-1, 4), coef)
    if spread:
        assert dec.probes > 0
    else:
        assert dec.probes == 0 and dec.radius_verdicts >= 1


def test_randomised_differential_open_paths():
    """~60 s (HB_STRESS_SECONDS), seeded: random shapes / arrival orders / edge-heavy inputs through the matrix-core and the integer-VALU
    kernels (scratch/stress_open_paths.py ran 248 k such opens in round 1; this is its bounded twin inside the suite),
    now including shapes that take the full-size matrix-core kernel (omega points, large powers)."""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, int("80" * 32, 16) % P, int("7f" * 32, 16) % P, int("ff00" * 16, 16) % P, 1 << 254, (1 << 254) - 1]
    rnd = random.Random(20260928)
    ctx = Context.get(P)
    as_np = lambda tns: tns.cpu().numpy().view(np.uint64)  # noqa: E731
    t_end = time.time() + float(os.environ.get("HB_STRESS_SECONDS", "60"))
    trials = wide = 0
    while time.time() < t_end or trials < 40:
        n = rnd.choice([4, 5, 7, 8, 13, 16, 17, 22, 31, 32, 33, 40, 47, 48, 49, 63, 64, 100, 128])
        t = rnd.randrange(0, min(n, 32))
        use_omega = rnd.random() < 0.25
        d = t + 1
        b = rnd.choice([1, d, d + 1, 16 * d, 16 * d + 1, 33 * d - 1, rnd.randrange(1, 3000)])
        c = (b + d - 1) // d
        order = list(range(n))
        rnd.shuffle(order)
        z, zc = order[:d], order[d : d + min(t, n - d)]
        op = BatchOpen(P, n, t, z=z, zc=zc, use_omega_powers=use_omega, max_shares=b)
        if not op.uses_matrix_cores():
            continue
        wide += use_omega or n ** t >= 127 * 256 ** 15
        shares = [rnd.choice(edge) if rnd.random() < rnd.choice([0.0, 0.1, 0.9]) else rnd.randrange(P) for _ in range(b)]
        sh = ctx.upload_ints(shares)
        enc_m = op.r1_encode(sh)
        op.set_matrix_cores(False)
        enc_v = op.r1_encode(sh)
        assert np.array_equal(as_np(enc_m), as_np(enc_v)), ("encode", n, t, b, use_omega)
        bad = None
        if zc and rnd.random() < 0.5:
            bad = enc_m.clone()
            bad[rnd.choice(zc) * c + rnd.randrange(c), rnd.randrange(4)] ^= 1 << rnd.randrange(60)
        outs = []
        for on in (True, False):
            op.set_matrix_cores(on)
            msg = op.r1_decode(enc_m, b)
            res = op.r2_decode(enc_m, b)
            assert op.ok(), ("validate", n, t, b, on, use_omega)
            assert ctx.download_ints(res) == shares, ("decode", n, t, b, on, use_omega)
            outs.append((as_np(msg).copy(), as_np(res).copy()))
            if bad is not None:
                op.r2_decode(bad, b)
                assert not op.ok(), ("corruption missed", n, t, b, on, use_omega)
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        trials += 1
        del op
    torch.cuda.synchronize()
    assert trials >= 40 and wide >= 5


def _table_recycling_trials(seconds, seed):
    """Plans created and destroyed in a tight loop on a context whose table cache holds ONE entry: every plan's tables are
    built at addresses freed a moment ago, and the FIRST launch over each fresh table is what is compared (round 2's stale-table
    defect only ever showed there: DESIGN section 9).  Returns (trials, failures) -- run in a subprocess so that HB_UPLOAD_MODE
    and the cap are read by a fresh library."""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    rnd = random.Random(seed)
    ctx = Context.get(P)
    as_np = lambda tns: tns.cpu().numpy().view(np.uint64)  # noqa: E731
    t_end = time.time() + seconds
    trials, failures = 0, []
    while time.time() < t_end:
        n = rnd.choice([16, 22, 31, 32, 48, 64, 64, 64, 100])
        t = rnd.randrange(3, min(n // 3 + 1, 32))
        use_omega = rnd.random() < 0.5
        d = t + 1
        b = rnd.choice([16 * d, 33 * d - 1, 300 * d, rnd.randrange(1, 6000)])
        c = (b + d - 1) // d
        order = list(range(n))
        rnd.shuffle(order)
        z, zc = order[:d], order[d : d + min(t, n - d)]
        op = BatchOpen(P, n, t, z=z, zc=zc, use_omega_powers=use_omega, max_shares=b)
        if not op.uses_matrix_cores():
            continue
        if op.uses_fused_validate():
            op.set_fused_validate(True)                      # the fused matrices too: built now, first launch below
        gen = torch.Generator(device="cuda")
        gen.manual_seed(rnd.randrange(1 << 30))
        sh = _rand(torch, b, gen)
        enc_first = as_np(op.r1_encode(sh)).copy()           # first launch over the freshly built encode table
        res_first = as_np(op.r2_decode(torch.from_numpy(enc_first.view(np.int64)).cuda(), b)).copy()
        ok_first = op.ok()
        enc_again = as_np(op.r1_encode(sh))
        res_again = as_np(op.r2_decode(torch.from_numpy(enc_first.view(np.int64)).cuda(), b))
        ok_again = op.ok()
        want = as_np(sh)
        if not (np.array_equal(enc_first, enc_again) and np.array_equal(res_first, want) and np.array_equal(res_again, want) and ok_first and ok_again):
            failures.append((n, t, b, use_omega, bool(np.array_equal(enc_first, enc_again)), bool(np.array_equal(res_first, want)), ok_first, ok_again))
        trials += 1
        del op
    torch.cuda.synchronize()
    return trials, failures


def _run_recycling_subprocess(seconds, seed, upload_mode=None):
    import json
    import subprocess
    import sys

    from conftest import REPO

    env = dict(os.environ, HB_CACHE_CAP="1", HB_PLAN_CACHE="0")
    env.pop("HB_UPLOAD_MODE", None)
    if upload_mode:
        env["HB_UPLOAD_MODE"] = upload_mode
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_full_size as T; "
            "tr, fl = T._table_recycling_trials(%r, %r); print('RESULT ' + json.dumps({'trials': tr, 'failures': fl}))"
            % (REPO, os.path.join(REPO, "tests"), seconds, seed))
    res = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=seconds + 600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_table_recycling_first_launch():
    """create / destroy plans in a loop with HB_CACHE_CAP=1 and compare every first launch (~25 s).  The
    production upload path (table images written by a copy KERNEL on the stream) must show no stale table."""
    out = _run_recycling_subprocess(25.0, 20260928)
    assert out["trials"] >= 200, out["trials"]
    assert not out["failures"], out["failures"][:5]


# ---------------------------------------------------------------------------------------------- narrow contexts
@pytest.mark.parametrize("p", [(1 << 64) - 59, 0xFFFFFFFF00000001, (1 << 61) - 1])
def test_narrow_context_every_entry_point(p):
    """north star "64/256-bit prime": a modulus below 2^64 gets the 1-limb context (8-byte elements, 3 x 29-bit digits)
    from Context.get, and every entry point of the drop-in on it equals the oracle."""
    from honeybadgermpc_amd import ntl
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder, wb_decode_batch

    ctx = Context.get(p)
    assert ctx.n_limbs == 1 and ctx.nbytes == 8 and ctx.empty(3).shape == (3, 1)
    rnd = random.Random(p % 9973)
    n, d, c = 32, 11, 300
    x = [rnd.randrange(1, p) for _ in range(n)]
    polys = [[rnd.randrange(p) for _ in range(d)] for _ in range(c)]
    ev = oracle.vandermonde_batch_evaluate(x, polys, p)
    assert ntl.vandermonde_batch_evaluate(x, polys, p) == ev
    zsel = rnd.sample(range(n), d)
    assert ntl.vandermonde_batch_interpolate([x[j] for j in zsel], [[row[j] for j in zsel] for row in ev], p) == polys
    # NTT (2-adicity: 2^64 - 59 has p - 1 = 2 * odd: order-2 transforms only; the other two primes take order 32)
    order = 32 if (p - 1) % 32 == 0 else 2
    g = next(a for a in range(2, 1000) if pow(a, (p - 1) // 2, p) != 1)
    omega = pow(g, (p - 1) // order, p)
    dd = min(d, order)
    co = [row[:dd] for row in polys]
    fev = oracle.fft_batch_evaluate(co, omega, p, order, order)
    assert ntl.fft_batch_evaluate(co, omega, p, order, order) == fev
    zs = rnd.sample(range(order), dd)
    assert ntl.fft_batch_interpolate(zs, [[row[j] for j in zs] for row in fev], omega, p, order) == co
    # Gao and Welch-Berlekamp with errors
    words = []
    for row in ev[:40]:
        row = list(row)
        for j in rnd.sample(range(n), rnd.randrange(0, (n - d) // 2 + 1)):
            row[j] = rnd.randrange(p)
        words.append(row)
    assert ntl.gao_interpolate_batch(x, words, d, p) == oracle.gao_interpolate_batch(x, words, d, p)
    xs = list(range(1, 17))
    wpolys = [[rnd.randrange(p) for _ in range(5)] for _ in range(12)]
    wrows = []
    for row in oracle.vandermonde_batch_evaluate(xs, wpolys, p):
        row = list(row)
        for j in rnd.sample(range(16), rnd.randrange(0, 7)):
            row[j] = rnd.randrange(p) if rnd.random() < 0.8 else None
        wrows.append(row)
    assert wb_decode_batch(xs, 5, wrows, p) == oracle.wb_decode_batch(xs, 5, wrows, p)
    assert ntl.sqrt_mod(pow(12345, 2, p), p) in (12345 % p, p - 12345 % p)
    # the open and the device decoder
    nn, t, b = 16, 5, 200
    dq = t + 1
    cq = (b + dq - 1) // dq
    xq = list(range(1, nn + 1))
    shares = [rnd.randrange(p) for _ in range(b)]
    p1 = [[rnd.randrange(p) for _ in range(dq)] for _ in range(cq)]
    p2 = [[rnd.randrange(p) for _ in range(dq)] for _ in range(cq)]
    e1, e2 = oracle.vandermonde_batch_evaluate(xq, p1, p), oracle.vandermonde_batch_evaluate(xq, p2, p)
    r1c = [[e1[k][j] for k in range(cq)] for j in range(nn)]
    r2c = [[e2[k][j] for k in range(cq)] for j in range(nn)]
    order_ = list(range(nn))
    rnd.shuffle(order_)
    z, zc = order_[:dq], order_[dq : dq + t]
    lim = lambda rows: oracle._limbs([v for r in rows for v in r], p)  # noqa: E731
    rc, o_r1, o_msg, o_res = oracle.batch_open_limbs(p, nn, dq, xq, oracle._limbs(shares, p), lim(r1c), lim(r2c), z, zc)
    assert rc == 0
    op = BatchOpen(p, nn, t, z=z, zc=zc, max_shares=b)
    assert not op.uses_matrix_cores()
    assert ctx.download_ints(op.r1_encode(ctx.upload_ints(shares))) == oracle._ints(o_r1)
    assert ctx.download_ints(op.r1_decode(ctx.upload_ints([v for col in r1c for v in col]), b)) == oracle._ints(o_msg)
    assert ctx.download_ints(op.r2_decode(ctx.upload_ints([v for col in r2c for v in col]), b)) == oracle._ints(o_res)
    assert op.ok()
    dec = DeviceIncrementalDecoder(p, nn, t, batch_size=cq)
    liar = order_[1]
    for idx in order_:
        col = list(r2c[idx])
        if idx == liar:
            col[3] = (col[3] + 1) % p
        dec.add(idx, col)
        if dec.done():
            break
    res, errs = dec.get_results()
    assert errs == {liar} and ctx.download_ints(res.reshape(-1, 1)) == [v for row in p2 for v in row]


@pytest.mark.parametrize("p", [(1 << 64) - 59, 0xFFFFFFFF00000001, (1 << 61) - 1, (1 << 41) + 27, (1 << 41) - 21, 1000003])
def test_narrow_matrix_core_kernel_equals_the_integer_kernel(p, monkeypatch):
    """k_mv64m (hb_narrow.hip: the 8-byte elements' mat-vec on the int8 matrix cores, round 6 -- the three launches of an open plan over a prime
    below 2^64) against the oracle and against k_mv64 (the integer-VALU kernel: the same plan with HB_NO_MFMA=1): row counts that are not
    multiples of 16 (odd numbers of row tiles: a pair's second tile is absent), one to three K-blocks of 8 terms, chunk counts that are not
    multiples of 16, a single chunk, plans at omega powers (full-size points: the matrix-core kernel takes those too), elements whose bytes
    sit on the edges of the windows' bias, and a lie in a compared row (both kernels must refuse it).  The primes either side of 2^41 are
    the first that takes the matrix-core kernel (three Montgomery steps bring a sum below 2 p from there on) and the last that does not."""
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    ctx = Context.get(p)
    rnd = random.Random(p % 7919 + 6)
    omega_ok = (p - 1) % 128 == 0
    edge = [0, p - 1, 0x8080808080808080 % p, 0x7f7f7f7f7f7f7f7f % p, 0x80 % p, 0xff00000000000000 % p]
    lim = lambda rows: oracle._limbs([v for r in rows for v in r], p)  # noqa: E731
    for nn, t, b, om in ((4, 1, 7, False), (16, 5, 200, False), (33, 7, 8, False), (40, 13, 3000, False), (64, 21, 22 * 1000 + 5, False), (70, 23, 999, False),
                         (64, 21, 4000, True), (24, 7, 129, True), (320, 21, 500, False)):     # (the last: an encode image beyond a launch's 64 KB of LDS -- k_mv64)
        if om and not omega_ok:
            continue
        dq = t + 1
        cq = (b + dq - 1) // dq
        shares = [rnd.choice(edge) if rnd.random() < 0.05 else rnd.randrange(p) for _ in range(b)]
        p1 = [[rnd.choice(edge) if rnd.random() < 0.05 else rnd.randrange(p) for _ in range(dq)] for _ in range(cq)]
        p2 = [[rnd.randrange(p) for _ in range(dq)] for _ in range(cq)]
        order_ = list(range(nn))
        rnd.shuffle(order_)
        z, zc = order_[:dq], order_[dq : dq + t]
        flat = lambda cc: ctx.upload_ints([v for col in cc for v in col])  # noqa: E731
        for hook in (None, "1"):
            if hook:
                set_hook(monkeypatch, "HB_NO_MFMA", hook)
            else:
                clear_hook(monkeypatch, "HB_NO_MFMA")
            op = BatchOpen(p, nn, t, z=z, zc=zc, use_omega_powers=om, max_shares=b)
            xq = op.x          # (omega powers: a prime whose seeded candidate root is not primitive draws again UNSEEDED, as the reference does -- the plan's own points)
            e1, e2 = oracle.vandermonde_batch_evaluate(xq, p1, p), oracle.vandermonde_batch_evaluate(xq, p2, p)
            r1c = [[e1[k][j] for k in range(cq)] for j in range(nn)]
            r2c = [[e2[k][j] for k in range(cq)] for j in range(nn)]
            rc, o_r1, o_msg, o_res = oracle.batch_open_limbs(p, nn, dq, xq, oracle._limbs(shares, p), lim(r1c), lim(r2c), z, zc)
            assert rc == 0
            lied = [list(col) for col in r2c]
            lied[zc[-1]][cq - 1] = (lied[zc[-1]][cq - 1] + 1) % p
            assert ctx.download_ints(op.r1_encode(ctx.upload_ints(shares))) == oracle._ints(o_r1), (nn, t, b, om, hook)
            assert ctx.download_ints(op.r1_decode(flat(r1c), b)) == oracle._ints(o_msg), (nn, t, b, om, hook)
            assert ctx.download_ints(op.r2_decode(flat(r2c), b)) == oracle._ints(o_res), (nn, t, b, om, hook)
            assert op.ok(), (nn, t, b, om, hook)
            op.r2_decode(flat(lied), b)
            assert not op.ok(), (nn, t, b, om, hook)


# ---------------------------------------------------------------------------------------------- bounded caches
def test_table_caches_are_bounded(monkeypatch):
    """the per-context caches are keyed by the arrival set of every asynchronous open.  With a cap of 16 entries,
    hundreds of different arrival sets leave at most the cap (+ the entries of the call in flight) resident, results stay
    right, the sorted-set key makes re-orderings of one set hit the same table, and cache_clear drops everything."""
    from honeybadgermpc_amd import ntl
    from honeybadgermpc_amd._capi import Context

    p = (1 << 255) - 19                      # a modulus no other test holds a context for: the cap is read at context creation
    set_hook(monkeypatch, "HB_CACHE_CAP", "16")
    Context._cache.pop((p, 0, 4), None)
    ctx = Context.get(p, 0)
    rnd = random.Random(8)
    n, d = 24, 6
    x = list(range(1, n + 1))
    polys = [[rnd.randrange(p) for _ in range(d)] for _ in range(4)]
    ev = oracle.vandermonde_batch_evaluate(x, polys, p)
    high = 0
    for _ in range(150):
        z = rnd.sample(range(n), d)
        assert ntl.vandermonde_batch_interpolate([x[j] for j in z], [[row[j] for j in z] for row in ev], p) == polys
        words = [list(row) for row in ev]
        sub = sorted(rnd.sample(range(n), 16))
        assert ntl.gao_interpolate_batch([x[j] for j in sub], [[row[j] for j in sub] for row in words], d, p)[0][0] == polys[0]
        high = max(high, ctx.cache_entries())
    assert high <= 16 + 8, high
    # one set, many orders: one table
    ctx.cache_clear()
    assert ctx.cache_entries() == 0
    zset = rnd.sample(range(n), d)
    for _ in range(20):
        rnd.shuffle(zset)
        assert ntl.vandermonde_batch_interpolate([x[j] for j in zset], [[row[j] for j in zset] for row in ev], p) == polys
    assert ctx.cache_entries() <= 1 + 20     # the table + at most one cached permutation per order
    tables = ctx.cache_entries()
    ctx.cache_clear()
    assert ctx.cache_entries() == 0 and tables >= 1


def test_one_context_shared_by_concurrent_threads(monkeypatch):
    """ctypes releases the GIL, so threads that share one Context (= one hb_ctx) really are inside the library at
    the same time.  Four threads create plans for different arrival sets, interpolate, Gao-decode and open concurrently with a
    table cap of 8 entries (cache_trim runs on almost every call and frees tables other threads looked up a moment ago);
    every result must equal the single-threaded answer.  The library serialises its table / cache / refcount sections
    per context (hb_common.hpp: HB_API_GUARD)."""
    import threading

    import torch

    from honeybadgermpc_amd import ntl
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    p = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141   # secp256k1's group order: a prime no other test holds a context for
    set_hook(monkeypatch, "HB_CACHE_CAP", "8")
    Context._cache.pop((p, 0, 4), None)
    ctx = Context.get(p, 0)
    n, t = 24, 5
    d = t + 1
    x = list(range(1, n + 1))
    rnd = random.Random(77)
    polys = [[rnd.randrange(p) for _ in range(d)] for _ in range(6)]
    ev = oracle.vandermonde_batch_evaluate(x, polys, p)                      # [6][n]
    c = 700
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    coef = _rand(torch, c * d, gen)                                            # < 2^253 < p: canonical
    V = BatchOpen(p, n, t, max_shares=c * d)
    cols = V.r1_encode(coef).clone()                                           # [n][c]: values of c polynomials at all points
    torch.cuda.synchronize()
    errors, rounds = [], 40

    def worker(seed):
        try:
            r = random.Random(seed)
            for _ in range(rounds):
                z = r.sample(range(n), d)
                zc = [j for j in r.sample(range(n), n) if j not in z][:t]
                got = ntl.vandermonde_batch_interpolate([x[j] for j in z], [[row[j] for j in z] for row in ev], p)
                if got != polys:
                    errors.append(("interpolate", seed, z))
                sub = sorted(r.sample(range(n), 16))
                g = ntl.gao_interpolate_batch([x[j] for j in sub], [[row[j] for j in sub] for row in ev], d, p)
                if [a for a, _ in g] != polys:
                    errors.append(("gao", seed, sub))
                op = BatchOpen(p, n, t, z=z, zc=zc, max_shares=c * d)
                res = op.r2_decode(cols, c * d)
                fine = op.ok()
                if not fine or not torch.equal(res, coef):
                    errors.append(("open", seed, z, fine))
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(("exception", seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
        assert not th.is_alive()
    assert not errors, errors[:5]
    ctx.cache_clear()


def test_open_plans_are_cached_per_thread(monkeypatch):
    """A plan costs ~2 ms to build; the device decoder takes its plans from a per-thread LRU keyed by everything that
    determines them, bounded by HB_PLAN_CACHE, and switched off by 0.  Two decoders fed the same arrival pattern share
    plans and still decide independently (the mismatch flag is read right after the launch it belongs to)."""
    import threading

    import torch

    from honeybadgermpc_amd import device
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import DeviceIncrementalDecoder, cached_batch_open

    ctx = Context.get(P)
    n, t, c = 16, 5, 40
    d = t + 1
    rnd = random.Random(4)
    polys = [[rnd.randrange(P) for _ in range(d)] for _ in range(c)]
    x = list(range(1, n + 1))
    ev = oracle.vandermonde_batch_evaluate(x, polys, P)
    cols = [ctx.upload_ints([ev[k][j] for k in range(c)]) for j in range(n)]
    device._plan_cache.plans.clear()
    a = cached_batch_open(P, n, t, list(range(d)), list(range(d, d + t)), max_shares=c * d)
    assert cached_batch_open(P, n, t, list(range(d)), list(range(d, d + t)), max_shares=c * d) is a
    assert cached_batch_open(P, n, t, list(range(1, d + 1)), [0], max_shares=c * d) is not a
    seen = {}
    th = threading.Thread(target=lambda: seen.setdefault("other", cached_batch_open(P, n, t, list(range(d)), list(range(d, d + t)), max_shares=c * d)))
    th.start(); th.join()
    assert seen["other"] is not a                   # never across threads
    set_hook(monkeypatch, "HB_PLAN_CACHE", "2")
    for s in range(5):
        cached_batch_open(P, n, t, list(range(s, s + d)), [], max_shares=c * d)
    assert len(device._plan_cache.plans) <= 2
    set_hook(monkeypatch, "HB_PLAN_CACHE", "0")
    assert cached_batch_open(P, n, t, list(range(d)), [], max_shares=c * d) is not cached_batch_open(P, n, t, list(range(d)), [], max_shares=c * d)
    monkeypatch.delenv("HB_PLAN_CACHE")
    # same arrival pattern twice, the second time with a liar: the shared plans must not leak the first run's verdict
    # (the plan-based decoder path: contexts / shapes the plan-free kernels do not take, or HB_NO_QUICK=1)
    set_hook(monkeypatch, "HB_NO_QUICK", "1")
    device._plan_cache.plans.clear()
    for liar in (None, 2):
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=c)
        for j in range(n):
            col = cols[j]
            if liar == j:
                col = col.clone(); col[:, 0] ^= 1
            dec.add(j, col)
            if dec.done():
                break
        coeffs, errs = dec.get_results()
        assert ctx.download_ints(coeffs.reshape(c * d, -1)) == [v for row in polys for v in row]
        assert errs == (set() if liar is None else {liar})
    assert len(device._plan_cache.plans) >= 1
    # the default, plan-free path builds no plan at all for the same two runs
    clear_hook(monkeypatch, "HB_NO_QUICK")
    device._plan_cache.plans.clear()
    for liar in (None, 2):
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=c)
        for j in range(n):
            col = cols[j]
            if liar == j:
                col = col.clone(); col[:, 0] ^= 1
            dec.add(j, col)
            if dec.done():
                break
        coeffs, errs = dec.get_results()
        assert ctx.download_ints(coeffs.reshape(c * d, -1)) == [v for row in polys for v in row]
        assert errs == (set() if liar is None else {liar}) and dec.quick_launches >= 1
    assert len(device._plan_cache.plans) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("p", [0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141, (1 << 256) - 189,
                               (1 << 256) - (1 << 32) - 977, (1 << 255) + 95, P])
def test_moduli_above_2_255_on_the_matrix_cores(p):
    """Round-3 finding: for p > 2^255 a Barrett remainder r < 2p can reach 2^256, and the matrix-core epilogues (k_mm8, k_mm8w,
    k_prescale_tab) decided r >= p from the eight packed words alone -- about one output in 10^4 came back as r - 2^256 + ... wrong
    (secp256k1's group order: 1 of 16 800 encode outputs).  Large seeded batches through both kernel families, encode / fused
    and unfused decode, plus an oracle subset."""
    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen

    if pow(2, p - 1, p) != 1:
        pytest.skip("not a prime")
    ctx = Context.get(p, 0)
    n, t = 24, 5
    d = t + 1
    c = 6000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(p % 1000003)
    coef = torch.randint(-(1 << 63), (1 << 63) - 1, (c * d, 4), dtype=torch.int64, device="cuda", generator=gen)
    coef[:, 3] &= (1 << (p.bit_length() - 193)) - 1          # below 2^(bits-1) <= p: canonical
    rnd = random.Random(9)
    as_np = lambda tns: tns.cpu().numpy().view(np.uint64)  # noqa: E731
    z = rnd.sample(range(n), d)
    zc = [j for j in range(n) if j not in z][:t]
    op = BatchOpen(p, n, t, z=z, zc=zc, max_shares=c * d)
    assert op.uses_matrix_cores()
    enc_m = op.r1_encode(coef)
    op.set_matrix_cores(False)
    enc_v = op.r1_encode(coef)
    assert np.array_equal(as_np(enc_m), as_np(enc_v)), "k_mm8 encode differs from the integer-VALU encode"
    x = list(range(1, n + 1))
    ints = ctx.download_ints(coef[: 40 * d])
    want = oracle.vandermonde_batch_evaluate(x, [ints[i * d : (i + 1) * d] for i in range(40)], p)
    got = ctx.download_ints(enc_v.view(n, c, 4)[:, :40, :].contiguous().view(-1, 4))
    assert got == [want[k][j] for j in range(n) for k in range(40)]
    for mc, fused in ((True, True), (True, False), (False, False)):
        op.set_matrix_cores(mc)
        op.set_fused_validate(fused)
        res = op.r2_decode(enc_v, c * d)
        msg = op.r1_decode(enc_v, c * d)
        assert op.ok(), (mc, fused)
        assert torch.equal(res, coef), (mc, fused)
        assert torch.equal(msg, coef.view(c, d, 4)[:, 0, :]), (mc, fused)


@pytest.mark.parametrize("always_void", [False, True])
def test_a_void_probe_launch_never_fails_a_decode(monkeypatch, always_void):
    """hb_probe_feed may answer HB_ERR_RETRY (a workgroup of a probe over several workgroups gave up waiting for another on a crowded chip: the
    launch is void, the probe reset).  The decoder feeds the whole list once more on the fewest workgroups; a second void launch sends the
    verdict to the batched Gao kernels (reference reed_solomon.py:151-186 either way).  Injected here: the first call (or every call) of
    hb_probe_feed resets the probe and reports a void launch.  Same coefficients, same liars, nothing raised out of add()."""
    import torch

    from honeybadgermpc_amd import device
    from honeybadgermpc_amd._capi import HB_ERR_RETRY, Context
    from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder

    n, t, c = 64, 21, 500
    d = t + 1
    ctx = Context.get(P)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5 + always_void)

    def rand(count):
        v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
        v[:, 3] &= (1 << 61) - 1
        return v

    coef = rand(c * d)
    cols = BatchOpen(P, n, t, max_shares=c * d).r1_encode(coef).view(n, c, 4).clone()
    rng = np.random.Generator(np.random.PCG64(77))
    liars = sorted(rng.choice(n, size=t, replace=False).tolist())
    for j in liars:
        cols[j] = rand(c)
    honest = [j for j in rng.permutation(n).tolist() if j not in liars]
    step = len(honest) // (t + 1)
    order = []
    for i, j in enumerate(liars):
        order += honest[i * step:(i + 1) * step] + [j]
    order += honest[t * step:]
    real_feed, calls = ctx.lib.hb_probe_feed, {"n": 0, "void": 0}

    def feed(h, *args):
        calls["n"] += 1
        if always_void or calls["n"] == 1:
            calls["void"] += 1
            assert ctx.lib.hb_probe_reset(h) == 0
            return HB_ERR_RETRY
        return real_feed(h, *args)

    device._probe_pool.idle.clear()                  # (a pooled probe narrowed by an earlier run would hide the narrowing)
    monkeypatch.setattr(ctx.lib, "hb_probe_feed", feed)
    dec = DeviceIncrementalDecoder(P, n, t, batch_size=c, columns=cols)
    for j in order:
        dec.add(j)
        if dec.done():
            break
    res, errs = dec.get_results()
    assert dec.done() and errs == set(liars) and torch.equal(res.reshape(-1, 4), coef)
    assert calls["void"] >= 1 and dec.probes > 0
    if always_void:
        assert calls["void"] == calls["n"]           # every verdict came from the batched decoder
    monkeypatch.undo()
    device._probe_pool.idle.clear()


def test_spread_liars_decode_while_another_stream_keeps_the_chip_busy():
    """The probe over several workgroups (n = 256: six of them talk through memory, with bounded waits) while a second stream runs encodes back to
    back on every CU: the decoder still names exactly the liars and returns the shared polynomials -- through the probe, its retry on fewer
    workgroups, or the batched decoder, whichever the crowding allows."""
    import threading

    import torch

    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder

    n, t = 256, 85
    d = t + 1
    c = 3000
    ctx = Context.get(P)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(91)

    def rand(count):
        v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
        v[:, 3] &= (1 << 61) - 1
        return v

    coef = rand(c * d)
    cols = BatchOpen(P, n, t, use_omega_powers=True, max_shares=c * d).r1_encode(coef).view(n, c, 4).clone()
    rng = np.random.Generator(np.random.PCG64(3))
    liars = sorted(rng.choice(n, size=t, replace=False).tolist())
    for j in liars:
        cols[j] = rand(c)
    honest = [j for j in rng.permutation(n).tolist() if j not in liars]
    step = len(honest) // (t + 1)
    order = []
    for i, j in enumerate(liars):
        order += honest[i * step:(i + 1) * step] + [j]
    order += honest[t * step:]
    # the noise: config 3's encode (persistent workgroups on every CU) back to back on another stream, from another thread
    noise_op = BatchOpen(P, 64, 21, max_shares=1 << 20)
    noise_in = rand(1 << 20)
    noise_out = ctx.empty(64 * noise_op.chunks(1 << 20))
    side = torch.cuda.Stream()
    stop = threading.Event()

    def noise():
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(20):
                    noise_op.r1_encode(noise_in, out=noise_out)
                side.synchronize()

    th = threading.Thread(target=noise)
    th.start()
    try:
        for rep in range(3):
            dec = DeviceIncrementalDecoder(P, n, t, batch_size=c, use_omega_powers=True, columns=cols)
            for j in order:
                dec.add(j)
                if dec.done():
                    break
            res, errs = dec.get_results()
            assert dec.done() and errs == set(liars), rep
            assert torch.equal(res.reshape(-1, 4), coef), rep
    finally:
        stop.set()
        th.join()
    torch.cuda.synchronize()
