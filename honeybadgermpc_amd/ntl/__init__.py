"""
Drop-in for the reference's NTL/Cython extension ``honeybadgermpc.ntl``
(reference: honeybadgermpc/ntl/__init__.py:1 re-exporting
honeybadgermpc/ntl/hbmpc_ntl_helpers.pyx:73-455).

Same names, positional orders, padding / truncation rules and error behaviour;
the arithmetic runs on the MI355X through libhbmpc_hip.so (include/hbmpc_hip.h).
Inputs are lists/tuples of non-negative Python ints, outputs lists of canonical
ints, exactly like the reference boundary.

Packed batches.  Marshalling Python ints costs ~60 ns each on either side of the call -- more than the arithmetic.
Every batched argument (`polynomials`, `data_list`, `coeffs`, `ys_list`, `ys`) therefore also accepts the kernels' own
layout, and the result comes back in the same kind:
    numpy.ndarray  uint64, shape (rows, width, limbs)      little-endian limbs, limbs = 4 (p < 2^256) or 1 (p < 2^64)
    torch.Tensor   int64,  shape (rows, width, limbs)      host or device; a device tensor never leaves HBM
Values are reduced on entry like list inputs (pyx:31-32; one elementwise launch, `hb_reduce`): words at or above p are fine.  The points / exponents (x, zs, omega) stay small lists.
"""
import ctypes

import numpy as np

from .._capi import HB_ERR_SINGULAR, HB_OK, Context, HbView, np_ptr

__all__ = [
    "lagrange_interpolate", "evaluate", "vandermonde_inverse", "InterpolationError",
    "vandermonde_batch_interpolate", "vandermonde_batch_evaluate", "fft", "partial_fft",
    "fft_batch_evaluate", "fft_interpolate", "fft_batch_interpolate", "gao_interpolate",
    "sqrt_mod", "sqrt_mod_batch", "SetNTLNumThreads", "AvailableNTLThreads", "SetNumThreads", "GetMaxThreads",
]


class InterpolationError(Exception):
    """reference hbmpc_ntl_helpers.pyx:135"""


def _require_list(v):
    if not isinstance(v, (list, tuple)):
        raise ValueError("Invalid arguments")  # pyx:61-62


def _flat_padded(rows, width):
    flat = []
    for r in rows:
        flat.extend(r)
        if len(r) < width:
            flat.extend([0] * (width - len(r)))  # short rows are zero padded (pyx:180-181, 232-233)
    return flat


def _rows(flat, width, count):
    return [flat[i * width : (i + 1) * width] for i in range(count)]


class _Batch:
    """a batched argument on the device: `dev` (rows * width, limbs) tensor + how the caller handed it over"""

    def __init__(self, ctx, data, pad_to=None, what="batch"):
        t = ctx.torch
        self.ctx = ctx
        if isinstance(data, np.ndarray) or isinstance(data, t.Tensor):
            if data.ndim != 3 or data.shape[2] != ctx.n_limbs:
                raise ValueError(f"{what}: packed batches have shape (rows, width, {ctx.n_limbs})")
            self.rows, self.width = int(data.shape[0]), int(data.shape[1])
            if isinstance(data, np.ndarray):
                if data.dtype != np.uint64:
                    raise ValueError(f"{what}: numpy batches must be uint64 limbs")
                self.kind = "numpy"
                dev = ctx.to_device(data.reshape(self.rows * self.width, ctx.n_limbs))
            else:
                if data.dtype != t.int64:
                    raise ValueError(f"{what}: torch batches must be int64 (u64 limbs viewed as int64)")
                self.kind = "torch-device" if data.is_cuda else "torch-host"
                dev = data.reshape(self.rows * self.width, ctx.n_limbs).contiguous().to(ctx.tdev)
            if pad_to is not None and pad_to != self.width:
                raise ValueError(f"{what}: packed rows must have length {pad_to}")
            # to_ZZ_p (pyx:31-32): whatever words the caller packed, the kernels see canonical residues.  A caller's own
            # device tensor is never written: it is reduced into a fresh buffer; host batches were copied already.
            red = ctx.empty(dev.shape[0]) if self.kind == "torch-device" else dev
            ctx.check(ctx.lib.hb_reduce(ctx.h, ctx.ptr(dev), ctx.ptr(red), dev.shape[0], None, ctx.stream()), "reduce")
            self.dev = red
            return
        self.kind = "list"
        self.rows = len(data)
        self.width = max([len(r) for r in data]) if pad_to is None else pad_to
        self.dev = ctx.upload_ints(_flat_padded(data, self.width)) if self.rows * self.width else ctx.empty(1)

    def result(self, out, width):
        """(rows * width, limbs) device tensor -> what the caller's kind of batch looks like"""
        ctx = self.ctx
        if self.kind == "list":
            return _rows(ctx.download_ints(out), width, self.rows)
        shaped = out.view(self.rows, width, ctx.n_limbs)
        if self.kind == "torch-device":
            return shaped
        if self.kind == "torch-host":
            return shaped.cpu()
        return shaped.cpu().numpy().view(np.uint64)


# ---------------------------------------------------------------------------
# Vandermonde path
# ---------------------------------------------------------------------------
def vandermonde_batch_evaluate(x, polynomials, modulus):
    """result[j][i] = sum_l polynomials[j][l] * x[i]^l  (pyx:199-244)."""
    _require_list(x)
    ctx = Context.get(modulus)
    batch = _Batch(ctx, polynomials, what="polynomials")
    n, k, d = len(x), batch.rows, batch.width
    dout = ctx.empty(k * n)
    rc = ctx.lib.hb_vandermonde_batch_evaluate(
        ctx.h, np_ptr(ctx.host_elems(x)), n, ctx.ptr(batch.dev), k, d, ctx.ptr(dout), ctx.stream()
    )
    ctx.check(rc, "vandermonde_batch_evaluate")
    return batch.result(dout, n)


def vandermonde_batch_interpolate(x, data_list, modulus):
    """polynomials[j] = coefficients (untrimmed, len(x) of them) of the P_j with
    P_j(x[i]) = data_list[j][i]; InterpolationError when V(x) is singular (pyx:139-197)."""
    ctx = Context.get(modulus)
    batch = _Batch(ctx, data_list, what="data_list")
    k, n_chunks = batch.width, batch.rows
    if k != len(x):
        # NTL would abort on the dimension mismatch (unpinned in the reference); be explicit
        raise ValueError("vandermonde_batch_interpolate: len(x) must equal the row length")
    dout = ctx.empty(n_chunks * k)
    rc = ctx.lib.hb_vandermonde_batch_interpolate(
        ctx.h, np_ptr(ctx.host_elems(x)), k, ctx.ptr(batch.dev), n_chunks, ctx.ptr(dout), ctx.stream()
    )
    if rc == HB_ERR_SINGULAR:
        raise InterpolationError("Interpolation failed")
    ctx.check(rc, "vandermonde_batch_interpolate")
    return batch.result(dout, k)


def vandermonde_inverse(x, modulus):
    """Legacy dump of V(x)^-1 in NTL's matrix text format (pyx:115-132; unused by the
    reference's own callers)."""
    ctx = Context.get(modulus)
    k = len(x)
    m = ctypes.c_void_p()
    rc = ctx.lib.hb_vand_inverse_create(ctx.h, np_ptr(ctx.host_elems(x)), k, ctypes.byref(m), ctx.stream())
    if rc == HB_ERR_SINGULAR:
        # NTL leaves the result unspecified when det == 0; an empty matrix is the honest answer
        return "[]"
    ctx.check(rc, "vandermonde_inverse")
    out = np.zeros((k * k, ctx.n_limbs), dtype=np.uint64)
    try:
        ctx.check(ctx.lib.hb_matrix_to_host(ctx.h, m, np_ptr(out), ctx.stream()), "matrix_to_host")
    finally:
        ctx.lib.hb_matrix_destroy(m)          # this call's handle; the context's cache keeps (and bounds) its own reference
    from .._capi import limbs_to_ints

    vals = limbs_to_ints(out, ctx.nbytes)
    body = "\n".join("[" + " ".join(str(v) for v in vals[i * k : (i + 1) * k]) + "]" for i in range(k))
    return "[" + body + "\n]"


def lagrange_interpolate(x, y, modulus):
    """Coefficients of the unique P with deg P < len(x), P(x[i]) = y[i], trimmed to
    deg(P)+1 entries; the zero polynomial is [] (pyx:73-99, rsdecode_impl.h:67-90)."""
    assert len(x) == len(y)
    if len(x) == 0:
        return []
    try:
        coeffs = vandermonde_batch_interpolate(x, [list(y)], modulus)[0]
    except InterpolationError:
        raise ValueError("lagrange_interpolate: repeated evaluation point")
    while coeffs and coeffs[-1] == 0:
        coeffs.pop()
    return coeffs


def evaluate(polynomial, x, modulus):
    """P(x) (pyx:101-113)."""
    if len(polynomial) == 0:
        return 0
    return vandermonde_batch_evaluate([x], [list(polynomial)], modulus)[0][0]


# ---------------------------------------------------------------------------
# FFT path
# ---------------------------------------------------------------------------
def fft_batch_evaluate(coeffs, omega, modulus, n, k):
    """Row-wise first k of the n-point transform a[i] = sum_j c[j] omega^(ij) (pyx:286-316)."""
    ctx = Context.get(modulus)
    if isinstance(coeffs, (list, tuple)):
        d0 = len(coeffs[0])
        for row in coeffs:
            if len(row) != d0:
                # the reference sizes every row from row 0 (pyx:295): ragged input is UB there
                raise ValueError("fft_batch_evaluate: all rows must have the same length")
    n, k = int(n), int(k)
    if k > n or n <= 0 or n & (n - 1):
        raise ValueError("fft_batch_evaluate: n must be a power of two and k <= n")
    batch = _Batch(ctx, coeffs, what="coeffs")
    batch_size, d = batch.rows, batch.width
    dout = ctx.empty(batch_size * k)
    rc = ctx.lib.hb_fft_batch_evaluate(
        ctx.h, np_ptr(ctx.host_elems([omega])), n, ctx.ptr(batch.dev), batch_size, d, k, ctx.ptr(dout), ctx.stream()
    )
    ctx.check(rc, "fft_batch_evaluate")
    return batch.result(dout, k)


def fft(coeffs, omega, modulus, n):
    """n-point transform; coefficient lists longer than n are truncated (rsdecode_impl.h:173)."""
    return fft_batch_evaluate([list(coeffs)], omega, modulus, n, n)[0]


def partial_fft(coeffs, omega, modulus, n, k):
    return fft_batch_evaluate([list(coeffs)], omega, modulus, n, k)[0]


def fft_batch_interpolate(zs, ys_list, omega, modulus, n):
    """Row-wise P (k = len(zs) untrimmed coefficients) with P(omega^zs[i]) = ys[i] (pyx:342-381)."""
    ctx = Context.get(modulus)
    k = len(zs)
    za = np.array([int(z) for z in zs], dtype=np.int32)
    if k and (za.min() < 0 or za.max() >= int(n)):
        raise ValueError("fft_batch_interpolate: zs must lie in [0, n)")
    if isinstance(ys_list, (list, tuple)):
        ys_list = [list(row[:k]) for row in ys_list]
    batch = _Batch(ctx, ys_list, pad_to=k, what="ys_list")
    n_chunks = batch.rows
    dout = ctx.empty(n_chunks * k)
    rc = ctx.lib.hb_fft_batch_interpolate(
        ctx.h, np_ptr(ctx.host_elems([omega])), int(n), np_ptr(za), k, ctx.ptr(batch.dev), n_chunks, ctx.ptr(dout), ctx.stream()
    )
    if rc == HB_ERR_SINGULAR:
        raise ValueError("fft_batch_interpolate: zs must be distinct")
    ctx.check(rc, "fft_batch_interpolate")
    return batch.result(dout, k)


def fft_interpolate(zs, ys, omega, modulus, n):
    return fft_batch_interpolate(zs, [list(ys)], omega, modulus, n)[0]


# ---------------------------------------------------------------------------
# Gao decoder
# ---------------------------------------------------------------------------
def gao_interpolate_batch(x, ys, k, modulus):
    """C codewords sharing the points x (no erasures) decoded in one launch.
    Returns a list of (coeffs, error_poly) | (None, None).  (Batched form of pyx:389-439.)"""
    ctx = Context.get(modulus)
    batch = _Batch(ctx, ys, pad_to=len(x), what="ys")
    n, c = len(x), batch.rows
    dys = batch.dev
    dco = ctx.empty(c * k)
    derr = ctx.empty(c * (n + 1))
    t = ctx.torch
    dlen = t.zeros(c, dtype=t.int32, device=ctx.tdev)
    dok = t.zeros(c, dtype=t.uint8, device=ctx.tdev)
    rc = ctx.lib.hb_gao_decode(
        ctx.h, np_ptr(ctx.host_elems(x)), n, int(k), ctx.ptr(dys), c,
        ctx.ptr(dco), ctx.ptr(derr), ctx.ptr(dlen), ctx.ptr(dok), ctx.stream(),
    )
    ctx.check(rc, "gao_interpolate")
    if batch.kind != "list":
        # packed: (coeffs (C, k, limbs), error polynomials (C, n + 1, limbs) zero beyond their lengths, lengths (C,), ok (C,))
        keep = t.arange(n + 1, device=ctx.tdev).unsqueeze(0) < dlen.unsqueeze(1)
        derr = derr.view(c, n + 1, ctx.n_limbs) * keep.unsqueeze(2)
        if batch.kind == "torch-device":
            return dco.view(c, k, ctx.n_limbs), derr, dlen, dok.bool()
        if batch.kind == "torch-host":
            return dco.view(c, k, ctx.n_limbs).cpu(), derr.cpu(), dlen.cpu(), dok.bool().cpu()
        return (dco.view(c, k, ctx.n_limbs).cpu().numpy().view(np.uint64), derr.cpu().numpy().view(np.uint64), dlen.cpu().numpy(), dok.bool().cpu().numpy())
    ok, lens = dok.cpu().tolist(), dlen.cpu().tolist()
    co, er = ctx.download_ints(dco), ctx.download_ints(derr)
    out = []
    for i in range(c):
        if ok[i]:
            out.append((co[i * k : (i + 1) * k], er[i * (n + 1) : i * (n + 1) + lens[i]]))
        else:
            out.append((None, None))
    return out


def gao_interpolate(x, y, k, modulus, z=None, omega=None, order=None, use_omega_powers=False):
    """Gao's RS decoder (pyx:389-439, rsdecode_impl.h:281-405): drop erasures (None),
    then (message coeffs [k], error-locator cofactor coeffs) or (None, None).

    With use_omega_powers the reference interpolates g1 by FNT instead of NTL's
    interpolate (rsdecode_impl.h:376); g1 is the unique interpolant either way, so one
    kernel serves both and z / omega / order only get validated here."""
    assert len(x) == len(y)
    keep = [i for i in range(len(y)) if y[i] is not None]
    xs = [x[i] for i in keep]
    ys = [y[i] for i in keep]
    if use_omega_powers is True:
        assert z is not None
        assert len([z[i] for i in keep]) == len(xs)
        assert omega is not None
        int(order)
    return gao_interpolate_batch(xs, [ys], int(k), modulus)[0]


# ---------------------------------------------------------------------------
# misc
# ---------------------------------------------------------------------------
def sqrt_mod_batch(values, modulus):
    """Square roots of a list of residues in one launch; ValueError if any is a non-residue."""
    ctx = Context.get(modulus)
    t = ctx.torch
    c = len(values)
    din = ctx.upload_ints(list(values))
    dout = ctx.empty(c)
    dok = t.zeros(c, dtype=t.uint8, device=ctx.tdev)
    ctx.check(ctx.lib.hb_sqrt_mod(ctx.h, ctx.ptr(din), c, ctx.ptr(dout), ctx.ptr(dok), ctx.stream()), "sqrt_mod")
    if not bool(dok.all().item()):
        raise ValueError("sqrt_mod: not a quadratic residue")
    return ctx.download_ints(dout)


def sqrt_mod(a, n):
    """Some r with r*r = a (mod n), n an odd prime (pyx:441-444, NTL SqrRootMod).  Which root is
    unpinned by the reference (tests/test_ntl.py:331-341 checks r^2 only).  Tonelli-Shanks on the GPU."""
    return sqrt_mod_batch([a], n)[0]


# Thread knobs exist only for API parity: the GPU launch geometry is not a user setting.
# AvailableNTLThreads() keeps returning what was last set because DecoderSelector's policy
# (reference reed_solomon.py:452-459) reads it.
_threads = 1


def SetNTLNumThreads(x):  # noqa: N802
    global _threads
    _threads = int(x)


def AvailableNTLThreads():  # noqa: N802
    return _threads


def SetNumThreads(n):  # noqa: N802
    SetNTLNumThreads(n)


def GetMaxThreads():  # noqa: N802
    import os

    return os.cpu_count() or 1
