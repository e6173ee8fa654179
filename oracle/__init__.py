"""
oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of ``oracle/hbmpc_oracle.c`` (a plain-C CPU restatement of the
reference's batch share-reconstruction arithmetic).  It exposes the same Python
names, positional orders, padding/truncation rules and error behaviour as the
reference's ``honeybadgermpc/ntl/__init__.py:1`` re-exports
(``hbmpc_ntl_helpers.pyx:73-455``) plus the pure-Python Welch-Berlekamp decoder
(``reed_solomon_wb.py:47-153``), so tests can compare
``honeybadgermpc_amd.ntl.f(...)`` with ``oracle.f(...)`` argument-for-argument.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``honeybadgermpc_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhbmpc_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "hbmpc_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_get_max_threads.restype = ctypes.c_int
    return _lib


class InterpolationError(Exception):
    """Mirrors hbmpc_ntl_helpers.pyx:135."""


# ---------------------------------------------------------------------------
# int <-> limb marshalling (the reference's wire format is little-endian bytes,
# hbmpc_ntl_helpers.pyx:20-29)
# ---------------------------------------------------------------------------
def _limbs(values, modulus):
    """list[int] -> (len, 4) uint64; values reduced mod p on entry (pyx:31-32);
    negative ints raise OverflowError exactly like int.to_bytes in pyx:20-22."""
    out = bytearray()
    for v in values:
        if v < 0:
            raise OverflowError("can't convert negative int to unsigned")
        if v >= modulus:
            v %= modulus
        out += v.to_bytes(32, "little")
    return np.frombuffer(bytes(out), dtype=np.uint64).reshape(-1, 4).copy()


def _ints(arr):
    b = arr.tobytes()
    return [int.from_bytes(b[i : i + 32], "little") for i in range(0, len(b), 32)]


def _p(modulus):
    if modulus >= 1 << 256:
        raise ValueError("oracle supports moduli below 2**256")
    return _limbs([modulus], modulus + 1)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check_list(v):
    if not isinstance(v, (list, tuple)):
        raise ValueError("Invalid arguments")  # pyx:61-62


def _pad_rows(rows, width):
    flat = []
    for r in rows:
        flat.extend(r)
        flat.extend([0] * (width - len(r)))
    return flat


# ---------------------------------------------------------------------------
# thread knobs (pyx:383-387, 446-455)
# ---------------------------------------------------------------------------
_ntl_threads = 1


def SetNTLNumThreads(x):  # noqa: N802
    global _ntl_threads
    _ntl_threads = int(x)


def AvailableNTLThreads():  # noqa: N802
    return _ntl_threads


def SetNumThreads(n):  # noqa: N802
    SetNTLNumThreads(n)
    lib().orc_set_num_threads(int(n))


def GetMaxThreads():  # noqa: N802
    return lib().orc_get_max_threads()


# ---------------------------------------------------------------------------
# entry points
# ---------------------------------------------------------------------------
def lagrange_interpolate(x, y, modulus):
    assert len(x) == len(y)
    n = len(x)
    xa, ya = _limbs(x, modulus), _limbs(y, modulus)
    out = np.zeros((max(n, 1), 4), dtype=np.uint64)
    out_len = ctypes.c_int(0)
    rc = lib().orc_lagrange_interpolate(
        _ptr(_p(modulus)), _ptr(xa), _ptr(ya), n, _ptr(out), ctypes.byref(out_len)
    )
    if rc != 0:
        raise ValueError("lagrange_interpolate: repeated evaluation point")
    return _ints(out[: out_len.value])


def evaluate(polynomial, x, modulus):
    ca = _limbs(polynomial, modulus)
    xa = _limbs([x], modulus)
    out = np.zeros((1, 4), dtype=np.uint64)
    lib().orc_evaluate(_ptr(_p(modulus)), _ptr(ca), len(polynomial), _ptr(xa), _ptr(out))
    return _ints(out)[0]


def vandermonde_inverse(x, modulus):
    k = len(x)
    xa = _limbs(x, modulus)
    out = np.zeros((k * k, 4), dtype=np.uint64)
    lib().orc_vandermonde_inverse(_ptr(_p(modulus)), _ptr(xa), k, _ptr(out))
    vals = _ints(out)
    rows = ["[" + " ".join(str(v) for v in vals[i * k : (i + 1) * k]) + "]" for i in range(k)]
    return "[" + "\n".join(rows) + "\n]"


def vandermonde_batch_interpolate(x, data_list, modulus):
    k = max([len(d) for d in data_list])
    n_chunks = len(data_list)
    if k != len(x):
        raise ValueError("vandermonde_batch_interpolate: len(x) != row length")
    xa = _limbs(x, modulus)
    da = _limbs(_pad_rows(data_list, k), modulus)
    out = np.zeros((n_chunks * k, 4), dtype=np.uint64)
    rc = lib().orc_vandermonde_batch_interpolate(
        _ptr(_p(modulus)), _ptr(xa), k, _ptr(da), ctypes.c_long(n_chunks), _ptr(out)
    )
    if rc == 1:
        raise InterpolationError("Interpolation failed")
    v = _ints(out)
    return [v[i * k : (i + 1) * k] for i in range(n_chunks)]


def vandermonde_batch_evaluate(x, polynomials, modulus):
    _check_list(x)
    n = len(x)
    k = len(polynomials)
    d = max([len(poly) for poly in polynomials])
    xa = _limbs(x, modulus)
    pa = _limbs(_pad_rows(polynomials, d), modulus)
    out = np.zeros((k * n, 4), dtype=np.uint64)
    lib().orc_vandermonde_batch_evaluate(
        _ptr(_p(modulus)), _ptr(xa), n, _ptr(pa), ctypes.c_long(k), d, _ptr(out)
    )
    v = _ints(out)
    return [v[i * n : (i + 1) * n] for i in range(k)]


def fft(coeffs, omega, modulus, n):
    return fft_batch_evaluate([coeffs], omega, modulus, n, n)[0]


def partial_fft(coeffs, omega, modulus, n, k):
    return fft_batch_evaluate([coeffs], omega, modulus, n, k)[0]


def fft_batch_evaluate(coeffs, omega, modulus, n, k):
    batch_size = len(coeffs)
    d = len(coeffs[0])
    for row in coeffs:
        if len(row) != d:
            raise ValueError("fft_batch_evaluate: ragged input (pyx:295 sizes every row from row 0)")
    ca = _limbs([c for row in coeffs for c in row], modulus)
    if d == 0:
        ca = np.zeros((1, 4), dtype=np.uint64)
    oa = _limbs([omega], modulus)
    out = np.zeros((max(batch_size * k, 1), 4), dtype=np.uint64)
    lib().orc_fft_batch_evaluate(
        _ptr(_p(modulus)), _ptr(oa), int(n), _ptr(ca), ctypes.c_long(batch_size), d, int(k), _ptr(out)
    )
    v = _ints(out[: batch_size * k])
    return [v[i * k : (i + 1) * k] for i in range(batch_size)]


def fft_interpolate(zs, ys, omega, modulus, n):
    return fft_batch_interpolate(zs, [ys], omega, modulus, n)[0]


def fft_batch_interpolate(zs, ys_list, omega, modulus, n):
    k = len(zs)
    n_chunks = len(ys_list)
    za = np.array([int(z) for z in zs], dtype=np.int32)
    ya = _limbs([y for row in ys_list for y in row[:k]], modulus)
    oa = _limbs([omega], modulus)
    out = np.zeros((max(n_chunks * k, 1), 4), dtype=np.uint64)
    rc = lib().orc_fft_batch_interpolate(
        _ptr(_p(modulus)), _ptr(oa), int(n), _ptr(za), k, _ptr(ya), ctypes.c_long(n_chunks), _ptr(out)
    )
    if rc != 0:
        raise ValueError("fft_batch_interpolate: zs must be distinct integers in [0, n)")
    v = _ints(out[: n_chunks * k])
    return [v[i * k : (i + 1) * k] for i in range(n_chunks)]


def gao_interpolate(x, y, k, modulus, z=None, omega=None, order=None, use_omega_powers=False):
    assert len(x) == len(y)
    is_null = [yi is None for yi in y]
    x = [x[i] for i in range(len(x)) if not is_null[i]]
    y = [y[i] for i in range(len(y)) if not is_null[i]]
    if z is not None:
        z = [z[i] for i in range(len(z)) if not is_null[i]]
    n = len(x)
    xa, ya = _limbs(x, modulus), _limbs(y, modulus)
    if use_omega_powers is True:
        assert z is not None
        assert len(z) == n
        assert omega is not None
        za = np.array([int(zi) for zi in z], dtype=np.int32)
        oa = _limbs([omega], modulus)
        order = int(order)
    else:
        za = np.zeros(max(n, 1), dtype=np.int32)
        oa = np.zeros((1, 4), dtype=np.uint64)
        order = 0
    res = np.zeros((max(k, 1), 4), dtype=np.uint64)
    err = np.zeros((n + 1, 4), dtype=np.uint64)
    err_len = np.zeros(1, dtype=np.int32)
    ok = np.zeros(1, dtype=np.uint8)
    lib().orc_gao_interpolate(
        _ptr(_p(modulus)), _ptr(xa), _ptr(za), n, int(k), _ptr(ya), ctypes.c_long(1),
        1 if use_omega_powers is True else 0, _ptr(oa), order,
        _ptr(res), _ptr(err), _ptr(err_len), _ptr(ok),
    )
    if ok[0]:
        return _ints(res[:k]), _ints(err[: int(err_len[0])])
    return None, None


def gao_interpolate_batch(x, ys, k, modulus, z=None, omega=None, order=None, use_omega_powers=False):
    """Oracle-only convenience: C codewords over the same points (no erasures)."""
    n, c = len(x), len(ys)
    xa = _limbs(x, modulus)
    ya = _limbs([v for row in ys for v in row], modulus)
    za = np.array([int(zi) for zi in (z if z is not None else [0] * n)], dtype=np.int32)
    oa = _limbs([omega if omega is not None else 0], modulus)
    res = np.zeros((c * k, 4), dtype=np.uint64)
    err = np.zeros((c * (n + 1), 4), dtype=np.uint64)
    err_len = np.zeros(c, dtype=np.int32)
    ok = np.zeros(c, dtype=np.uint8)
    lib().orc_gao_interpolate(
        _ptr(_p(modulus)), _ptr(xa), _ptr(za), n, int(k), _ptr(ya), ctypes.c_long(c),
        1 if use_omega_powers else 0, _ptr(oa), int(order or 0),
        _ptr(res), _ptr(err), _ptr(err_len), _ptr(ok),
    )
    rv, ev = _ints(res), _ints(err)
    out = []
    for i in range(c):
        if ok[i]:
            out.append((rv[i * k : (i + 1) * k], ev[i * (n + 1) : i * (n + 1) + int(err_len[i])]))
        else:
            out.append((None, None))
    return out


def sqrt_mod(a, n):
    out = np.zeros((1, 4), dtype=np.uint64)
    rc = lib().orc_sqrt_mod(_ptr(_p(n)), _ptr(_limbs([a], n)), _ptr(out))
    if rc != 0:
        raise ValueError("sqrt_mod: not a quadratic residue")
    return _ints(out)[0]


# ---------------------------------------------------------------------------
# Welch-Berlekamp (reed_solomon_wb.py:129-151), values are plain ints here
# ---------------------------------------------------------------------------
WB_MESSAGES = {1: "found no divisors!", 2: "No solution"}


def wb_decode_batch(x, k, rows, modulus):
    """rows: list of length-n lists with None for erasures.
    Returns list of (coeffs | None, status) with coeffs stripped of trailing zeros."""
    n, c = len(x), len(rows)
    xa = _limbs(x, modulus)
    ya = _limbs([0 if v is None else v for row in rows for v in row], modulus)
    present = np.array([0 if v is None else 1 for row in rows for v in row], dtype=np.uint8)
    out = np.zeros((c * n, 4), dtype=np.uint64)
    out_len = np.zeros(c, dtype=np.int32)
    status = np.zeros(c, dtype=np.int32)
    lib().orc_wb_decode(
        _ptr(_p(modulus)), _ptr(xa), n, int(k), _ptr(ya), _ptr(present), ctypes.c_long(c),
        _ptr(out), _ptr(out_len), _ptr(status),
    )
    vals = _ints(out)
    res = []
    for i in range(c):
        if status[i] == 0:
            res.append((vals[i * n : i * n + int(out_len[i])], 0))
        else:
            res.append((None, int(status[i])))
    return res


def wb_decode(x, k, encoded, modulus):
    """One codeword; raises like the reference's decoder closure does."""
    coeffs, status = wb_decode_batch(x, k, [encoded], modulus)[0]
    if status == 3:
        raise AssertionError("2 * t + 1 + c <= n")  # reed_solomon_wb.py:132
    if status == 1:
        raise ValueError("found no divisors!")  # reed_solomon_wb.py:127
    if status == 2:
        raise Exception("No solution")  # reed_solomon_wb.py:245
    return coeffs


# ---------------------------------------------------------------------------
# whole fault-free per-party open on packed limb arrays (cpu_baseline workload)
# ---------------------------------------------------------------------------
def batch_open_limbs(modulus, n, d, x, shares, r1_cols, r2_cols, z, zc, use_fft=False, omega=0, order=0, out=None):
    """All array arguments are (count, 4) uint64 canonical limb arrays.
    Returns (rc, r1_out[n*C], r2_msg[C], result[B]).  `out` = a previous return's three arrays to reuse."""
    b = shares.shape[0]
    c = (b + d - 1) // d
    xa = _limbs(x, modulus)
    oa = _limbs([omega], modulus)
    za = np.array(z, dtype=np.int32)
    zca = np.array(list(zc) if len(zc) else [0], dtype=np.int32)
    if out is not None:
        r1_out, r2_msg, result = out
    else:
        r1_out = np.zeros((n * c, 4), dtype=np.uint64)
        r2_msg = np.zeros((c, 4), dtype=np.uint64)
        result = np.zeros((b, 4), dtype=np.uint64)
    rc = lib().orc_batch_open(
        _ptr(_p(modulus)), n, d, 1 if use_fft else 0, _ptr(oa), int(order), _ptr(xa),
        _ptr(np.ascontiguousarray(shares)), ctypes.c_long(b),
        _ptr(np.ascontiguousarray(r1_cols)), _ptr(np.ascontiguousarray(r2_cols)),
        _ptr(za), _ptr(zca), len(zc), _ptr(r1_out), _ptr(r2_msg), _ptr(result),
    )
    return rc, r1_out, r2_msg, result


def batch_open_u64(modulus, n, d, x, shares, r1_cols, r2_cols, z, zc, out=None):
    """The same open for a word-size prime (p < 2^64), elements as uint64 arrays: orc_batch_open_u64 (CPU baseline / checker of bench.py's
    cfg3-p64 workload).  Returns (rc, r1_out[n*C], r2_msg[C], result[B])."""
    b = shares.shape[0]
    c = (b + d - 1) // d
    xa = np.array([v % modulus for v in x], dtype=np.uint64)
    za = np.array(z, dtype=np.int32)
    zca = np.array(list(zc) if len(zc) else [0], dtype=np.int32)
    if out is not None:
        r1_out, r2_msg, result = out
    else:
        r1_out, r2_msg, result = np.zeros(n * c, dtype=np.uint64), np.zeros(c, dtype=np.uint64), np.zeros(b, dtype=np.uint64)
    fn = lib().orc_batch_open_u64
    fn.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    rc = fn(int(modulus), n, d, _ptr(xa), _ptr(np.ascontiguousarray(shares, dtype=np.uint64)), b, _ptr(np.ascontiguousarray(r1_cols, dtype=np.uint64)),
            _ptr(np.ascontiguousarray(r2_cols, dtype=np.uint64)), _ptr(za), _ptr(zca), len(zc), _ptr(r1_out), _ptr(r2_msg), _ptr(result))
    return rc, r1_out, r2_msg, result
