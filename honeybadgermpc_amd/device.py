"""
Device-resident bulk API: the same operations as honeybadgermpc_amd.ntl /
reed_solomon, but on tensors that stay in HBM (no Python-int marshalling).

Element layout: int64 tensor of shape (count, 4) holding little-endian 4 x uint64
limbs of canonical residues -- the layout of the C ABI (include/hbmpc_hip.h).
"""
import ctypes

import numpy as np

from ._capi import HB_ERR_MISMATCH, HB_OK, Context, np_ptr


def wb_decode_batch(x, k, rows, modulus):
    """Welch-Berlekamp over a batch of codewords sharing the points x.
    rows: lists of n ints (None = erasure).  Returns [(coeffs | None, status)] with coeffs
    stripped of trailing zeros; status 0 ok, 1 "found no divisors!", 2 "No solution",
    3 too few points (reference reed_solomon_wb.py:129-151)."""
    ctx = Context.get(modulus)
    t = ctx.torch
    n, c = len(x), len(rows)
    ys = ctx.upload_ints([0 if v is None else v for row in rows for v in row])
    present = t.tensor([0 if v is None else 1 for row in rows for v in row], dtype=t.uint8, device=ctx.tdev)
    out = ctx.empty(c * k)
    out_len = t.zeros(c, dtype=t.int32, device=ctx.tdev)
    status = t.zeros(c, dtype=t.int32, device=ctx.tdev)
    rc = ctx.lib.hb_wb_decode(
        ctx.h, np_ptr(ctx.host_elems(x)), n, int(k), ctx.ptr(ys), ctx.ptr(present), c,
        ctx.ptr(out), ctx.ptr(out_len), ctx.ptr(status), ctx.stream(),
    )
    ctx.check(rc, "wb_decode")
    vals = ctx.download_ints(out)
    lens, st = out_len.cpu().tolist(), status.cpu().tolist()
    res = []
    for i in range(c):
        if st[i] == 0:
            res.append((vals[i * k : i * k + lens[i]], 0))
        else:
            res.append((None, st[i]))
    return res


class BatchOpen:
    """One party's fault-free batch open on device tensors (C ABI hb_open_*).

    Mirrors the compute of batch_reconstruct (reference batch_reconstruction.py:158-227):
        r1 = op.r1_encode(shares)            # [n][C] party-major: row j goes to party j
        ... exchange ...                     # r1_cols[j] = what party j sent us
        r2_msg = op.r1_decode(r1_cols)       # [C]: broadcast to everyone
        ... exchange ...
        result = op.r2_decode(r2_cols)       # [B]
        op.check()                           # raises if a validated column disagreed
    z = the d party indices whose columns are decoded (first arrivals),
    zc = the later arrivals validated against the re-encoded guess.
    """

    def __init__(self, modulus, n, t, z=None, zc=None, use_omega_powers=False, degree=None, max_shares=1 << 20, device=None):
        from .field import GF
        from .polynomial import EvalPoint

        self.ctx = ctx = Context.get(modulus, device)
        self.n, self.t = n, t
        self.d = (t if degree is None else degree) + 1
        d = self.d
        point = EvalPoint(GF(modulus), n, use_omega_powers=use_omega_powers)
        self.x = [point(i).value for i in range(n)]
        self.z = list(range(d)) if z is None else list(z)
        # optimistic path finishes once degree+1+t columns agree (reference reed_solomon.py:302-303,328-330)
        self.zc = [i for i in range(n) if i not in self.z][: t] if zc is None else list(zc)
        assert len(self.z) == d
        self.max_shares = int(max_shares)
        za = np.array(self.z, dtype=np.int32)
        zca = np.array(self.zc if self.zc else [0], dtype=np.int32)
        omega = point.omega.value if use_omega_powers else 0
        h = ctypes.c_void_p()
        rc = ctx.lib.hb_open_plan_create(
            ctx.h, n, d, 1 if use_omega_powers else 0, np_ptr(ctx.host_elems(self.x)),
            np_ptr(ctx.host_elems([omega])), int(point.order), np_ptr(za), np_ptr(zca), len(self.zc),
            self.max_shares, ctypes.byref(h), ctx.stream(),
        )
        ctx.check(rc, "hb_open_plan_create")
        self.h = h

    VALIDATE_ARRIVED_ONLY = 1

    def set_validate_arrived_only(self, on):
        """Re-encode only the output tiles that contain a compared (later-arrived) row instead of
        all n rows as the reference's encode_batch does (reed_solomon.py:313).  Same decision."""
        self.ctx.check(self.ctx.lib.hb_open_plan_set_option(self.h, self.VALIDATE_ARRIVED_ONLY, 1 if on else 0), "set_option")

    MATRIX_CORES = 2

    def set_matrix_cores(self, on):
        """Allow (default) or forbid the int8 matrix-core kernels for the encode and the validating
        re-encode; results are bit-identical either way."""
        self.ctx.check(self.ctx.lib.hb_open_plan_set_option(self.h, self.MATRIX_CORES, 1 if on else 0), "set_option")

    def uses_matrix_cores(self):
        """True when this plan's encode / validation run on the matrix cores (shapes qualify and not disabled)."""
        v = ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.hb_open_plan_get_option(self.h, self.MATRIX_CORES, ctypes.byref(v)), "get_option")
        return bool(v.value)

    def chunks(self, b):
        return (b + self.d - 1) // self.d

    def r1_encode(self, shares, out=None):
        b = shares.shape[0]
        c = self.chunks(b)
        if out is None:
            out = self.ctx.empty(self.n * c)
        rc = self.ctx.lib.hb_open_r1_encode(self.h, self.ctx.ptr(shares), b, self.ctx.ptr(out), self.ctx.stream())
        self.ctx.check(rc, "hb_open_r1_encode")
        return out

    def r1_decode(self, r1_cols, b, out=None):
        if out is None:
            out = self.ctx.empty(self.chunks(b))
        rc = self.ctx.lib.hb_open_r1_decode(self.h, self.ctx.ptr(r1_cols), b, self.ctx.ptr(out), self.ctx.stream())
        self.ctx.check(rc, "hb_open_r1_decode")
        return out

    def r2_decode(self, r2_cols, b, out=None):
        if out is None:
            out = self.ctx.empty(b)
        rc = self.ctx.lib.hb_open_r2_decode(self.h, self.ctx.ptr(r2_cols), b, self.ctx.ptr(out), self.ctx.stream())
        self.ctx.check(rc, "hb_open_r2_decode")
        return out

    def ok(self):
        """Synchronise; True when every validated column matched the guess."""
        rc = self.ctx.lib.hb_open_status(self.h, self.ctx.stream())
        if rc == HB_OK:
            return True
        if rc == HB_ERR_MISMATCH:
            return False
        self.ctx.check(rc, "hb_open_status")

    def __del__(self):
        try:
            self.ctx.lib.hb_open_plan_destroy(self.h)
        except Exception:
            pass


class BatchOpenPipeline:
    """Several independent opens in flight: `depth` plans, each on its own stream, used round-robin.

    A party opens many share arrays concurrently (Mpc.open_share_array under asyncio, reference mpc.py:101-219).
    Kernels of different launches overlap where one open's kernels leave the GPU idle (tails, the
    elementwise pass), so two opens in flight reconstruct about 20 % more shares per second than one at a
    time (bench.py: detail.shares_per_s_per_gpu_two_opens_in_flight).  Each open is computed exactly as by
    BatchOpen; only their scheduling changes.

        pipe = BatchOpenPipeline(p, n, t, z=z, zc=zc, max_shares=B)
        lane = pipe.next()                       # a BatchOpen bound to its own stream
        with lane.on_stream():
            r1 = lane.op.r1_encode(shares) ...   # same calls as BatchOpen
        pipe.ok()                                # synchronises every lane
    """

    class Lane:
        def __init__(self, op, stream, torch):
            self.op, self.stream, self._torch = op, stream, torch

        def on_stream(self):
            return self._torch.cuda.stream(self.stream)

    def __init__(self, modulus, n, t, depth=2, **kw):
        import torch

        self.torch = torch
        self.lanes = []
        for _ in range(max(1, int(depth))):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                self.lanes.append(self.Lane(BatchOpen(modulus, n, t, **kw), st, torch))
        torch.cuda.synchronize()
        self._next = 0

    def next(self):
        lane = self.lanes[self._next]
        self._next = (self._next + 1) % len(self.lanes)
        return lane

    def ok(self):
        """Synchronise all lanes; True when no lane saw a validation mismatch."""
        res = True
        for lane in self.lanes:
            with lane.on_stream():
                res = lane.op.ok() and res
        return res
