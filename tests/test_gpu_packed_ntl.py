"""The ntl drop-in with packed batches (numpy uint64 / torch int64 limbs, host or device): same values as the list
boundary, no per-int marshalling."""
import random

import numpy as np
import pytest

import oracle
from conftest import BLS as P

pytestmark = pytest.mark.gpu


def _limbs3(rows):
    flat = oracle._limbs([v for r in rows for v in r], P)
    return flat.reshape(len(rows), len(rows[0]), 4)


def _ints3(arr):
    a = np.ascontiguousarray(arr)
    vals = oracle._ints(a.reshape(-1, 4))
    return [vals[i * a.shape[1] : (i + 1) * a.shape[1]] for i in range(a.shape[0])]


@pytest.mark.parametrize("kind", ["numpy", "torch-host", "torch-device"])
def test_packed_batches_equal_list_boundary(kind):
    import torch

    from honeybadgermpc_amd import ntl
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import get_omega

    rnd = random.Random(3)
    n, d, c = 16, 6, 37
    x = list(range(1, n + 1))
    polys = [[rnd.randrange(P) for _ in range(d)] for _ in range(c)]

    def give(rows):
        a = _limbs3(rows)
        if kind == "numpy":
            return a
        t = torch.from_numpy(a.view(np.int64).copy())
        return t.cuda() if kind == "torch-device" else t

    def take(res):
        if kind == "numpy":
            assert isinstance(res, np.ndarray) and res.dtype == np.uint64
            return _ints3(res)
        assert isinstance(res, torch.Tensor) and res.is_cuda == (kind == "torch-device")
        return _ints3(res.cpu().numpy().view(np.uint64))

    want = oracle.vandermonde_batch_evaluate(x, polys, P)
    assert ntl.vandermonde_batch_evaluate(x, polys, P) == want                       # the list boundary itself
    assert take(ntl.vandermonde_batch_evaluate(x, give(polys), P)) == want
    z = rnd.sample(range(n), d)
    data = [[row[j] for j in z] for row in want]
    assert take(ntl.vandermonde_batch_interpolate([x[j] for j in z], give(data), P)) == polys
    omega = get_omega(GF(P), n, seed=0).value
    ev = oracle.fft_batch_evaluate(polys, omega, P, n, n)
    assert take(ntl.fft_batch_evaluate(give(polys), omega, P, n, n)) == ev
    zs = rnd.sample(range(n), d)
    assert take(ntl.fft_batch_interpolate(zs, give([[row[j] for j in zs] for row in ev]), omega, P, n)) == polys
    # Gao on packed codewords: two errors per codeword, all decoded
    bad = [list(row) for row in want]
    for row in bad:
        for j in rnd.sample(range(n), 2):
            row[j] = rnd.randrange(P)
    co, err, ln, ok = ntl.gao_interpolate_batch(x, give(bad), d, P)
    assert bool(ok.all()) and take(co) == polys
    listed = ntl.gao_interpolate_batch(x, bad, d, P)
    errs = _ints3(err.cpu().numpy().view(np.uint64) if hasattr(err, "cpu") else err)
    lens = [int(v) for v in (ln.cpu().tolist() if hasattr(ln, "cpu") else ln.tolist())]
    assert [errs[i][: lens[i]] for i in range(c)] == [e for _, e in listed]
    with pytest.raises(ValueError):
        ntl.vandermonde_batch_evaluate(x, np.zeros((3, 4, 2), dtype=np.uint64), P)


@pytest.mark.parametrize("kind", ["numpy", "torch-device"])
def test_packed_words_at_or_above_p_mean_their_residues(kind):
    """the reference reduces every value that enters its boundary (to_ZZ_p, pyx:31-32).  Packed batches whose
    words lie in [p, 2^256) -- p itself, 2p, 2^256 - 1, x + p -- must give what the list boundary gives for the same integers,
    on every batched entry point (small-entry and full-size kernels), and the caller's device tensor is left untouched."""
    import torch

    from honeybadgermpc_amd import ntl
    from honeybadgermpc_amd._capi import Context
    from honeybadgermpc_amd.field import GF
    from honeybadgermpc_amd.polynomial import get_omega

    rnd = random.Random(31)
    n, d, c = 16, 6, 300                      # >= 256 chunks: the matrix-core kernels take it
    x = list(range(1, n + 1))
    top = (1 << 256) - 1
    special = [P, P + 1, 2 * P, 2 * P + 5, top, top - 1, 0, 1, P - 1]

    def raw_value():
        r = rnd.random()
        if r < 0.3:
            return rnd.choice(special)
        if r < 0.7:
            return rnd.randrange(P, 1 << 256)
        return rnd.randrange(P)

    raw = [[raw_value() for _ in range(d)] for _ in range(c)]
    red = [[v % P for v in row] for row in raw]

    def pack_raw(rows):
        flat = np.zeros((len(rows) * len(rows[0]), 4), dtype=np.uint64)
        for i, v in enumerate(v for r in rows for v in r):
            for q in range(4):
                flat[i, q] = (v >> (64 * q)) & ((1 << 64) - 1)
        a = flat.reshape(len(rows), len(rows[0]), 4)
        if kind == "numpy":
            return a
        return torch.from_numpy(a.view(np.int64).copy()).cuda()

    def take(res):
        return _ints3(res if kind == "numpy" else res.cpu().numpy().view(np.uint64))

    given = pack_raw(raw)
    before = given.copy() if kind == "numpy" else given.clone()
    assert take(ntl.vandermonde_batch_evaluate(x, given, P)) == oracle.vandermonde_batch_evaluate(x, red, P)
    assert (given == before).all()
    z = rnd.sample(range(n), d)
    assert take(ntl.vandermonde_batch_interpolate([x[j] for j in z], pack_raw(raw), P)) == oracle.vandermonde_batch_interpolate([x[j] for j in z], red, P)
    omega = get_omega(GF(P), n, seed=0).value
    assert take(ntl.fft_batch_evaluate(pack_raw(raw), omega, P, n, n)) == oracle.fft_batch_evaluate(red, omega, P, n, n)
    zs = rnd.sample(range(n), d)
    assert take(ntl.fft_batch_interpolate(zs, pack_raw(raw), omega, P, n)) == oracle.fft_batch_interpolate(zs, red, omega, P, n)
    # hb_reduce itself, narrow context included
    ctx = Context.get(P)
    t = pack_raw(raw) if kind != "numpy" else torch.from_numpy(pack_raw(raw).view(np.int64).copy()).cuda()
    flat = t.reshape(c * d, 4).contiguous()
    assert ctx.download_ints(ctx.reduce_(flat)) == [v for r in red for v in r]
    q = (1 << 61) - 1
    ctxn = Context.get(q)
    vals = [rnd.choice([q, q + 1, 2 * q, (1 << 64) - 1, rnd.randrange(1 << 64)]) for _ in range(500)]
    tn = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in vals], dtype=torch.int64, device="cuda").reshape(-1, 1)
    assert ctxn.download_ints(ctxn.reduce_(tn)) == [v % q for v in vals]
