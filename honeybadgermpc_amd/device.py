"""
Device-resident bulk API: the same operations as honeybadgermpc_amd.ntl /
reed_solomon, but on tensors that stay in HBM (no Python-int marshalling).

Element layout: int64 tensor of shape (count, 4) holding little-endian 4 x uint64
limbs of canonical residues -- the layout of the C ABI (include/hbmpc_hip.h).
"""
import ctypes
import os
import threading
import weakref
from collections import OrderedDict

import numpy as np

from ._capi import HB_DEC_DISAGREE, HB_DEC_DONE, HB_DEC_OPT_BESIDE, HB_DEC_OPT_DEFER, HB_DEC_PENDING, HB_DEC_UNSUPPORTED, HB_ERR_MISMATCH, HB_ERR_RETRY, HB_ERR_UNSUPPORTED, HB_OK, Context, _marshal, np_ptr


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

    FUSED_VALIDATE = 3

    def set_fused_validate(self, on):
        """Allow (default) or forbid the one-launch decode + validate (include/hbmpc_hip.h, HB_OPEN_OPT_FUSED_VALIDATE); off = decode,
        re-encode all n points, compare.  on = "wide": the full-size kernel (hb_mfma_wide.hip) even where the small-entry kernel
        with the division inside (hb_mfma_fused.hip: points that are small integers) would be the default."""
        v = 2 if on == "wide" else (1 if on else 0)
        self.ctx.check(self.ctx.lib.hb_open_plan_set_option(self.h, self.FUSED_VALIDATE, v), "set_option")

    def fused_validate_kernel(self):
        """which kernel decodes + validates in one launch for this plan: "small", "wide" or None"""
        v = ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.hb_open_plan_get_option(self.h, self.FUSED_VALIDATE, ctypes.byref(v)), "get_option")
        return {0: None, 3: "small"}.get(v.value, "wide")

    def uses_fused_validate(self):
        v = ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.hb_open_plan_get_option(self.h, self.FUSED_VALIDATE, ctypes.byref(v)), "get_option")
        return bool(v.value)

    def uses_matrix_cores(self):
        """True when this plan's encode / validation run on the matrix cores (shapes qualify and not disabled)."""
        v = ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.hb_open_plan_get_option(self.h, self.MATRIX_CORES, ctypes.byref(v)), "get_option")
        return bool(v.value)

    def chunks(self, b):
        return (b + self.d - 1) // self.d

    def _check_batch(self, b):
        if b > self.max_shares:
            raise ValueError(f"{b} shares exceed this plan's max_shares = {self.max_shares}")

    def r1_encode(self, shares, out=None):
        shares = self.ctx.elems(shares, what="shares")
        b = shares.shape[0]
        self._check_batch(b)
        c = self.chunks(b)
        if out is None:
            out = self.ctx.empty(self.n * c)
        else:
            out = self.ctx.elems(out, self.n * c, what="out")
        rc = self.ctx.lib.hb_open_r1_encode(self.h, self.ctx.ptr(shares), b, self.ctx.ptr(out), self.ctx.stream())
        self.ctx.check(rc, "hb_open_r1_encode")
        return out

    def r1_decode(self, r1_cols, b, out=None):
        self._check_batch(b)
        r1_cols = self.ctx.elems(r1_cols, self.n * self.chunks(b), what="r1_cols")
        if out is None:
            out = self.ctx.empty(self.chunks(b))
        else:
            out = self.ctx.elems(out, self.chunks(b), what="out")
        rc = self.ctx.lib.hb_open_r1_decode(self.h, self.ctx.ptr(r1_cols), b, self.ctx.ptr(out), self.ctx.stream())
        self.ctx.check(rc, "hb_open_r1_decode")
        return out

    def r2_decode(self, r2_cols, b, out=None):
        self._check_batch(b)
        r2_cols = self.ctx.elems(r2_cols, self.n * self.chunks(b), what="r2_cols")
        if out is None:
            out = self.ctx.empty(b)
        else:
            out = self.ctx.elems(out, b, what="out")
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


class _PlanCache(threading.local):
    """Open plans of this thread, most recently used last.  A plan costs ~2 ms to create (tables, images, a handful of
    synchronisations) -- as much as decoding 2^20 shares -- and consecutive opens of an MPC program see the same few arrival
    patterns; a plan is immutable apart from its mismatch flag, which its user reads right after the launch it belongs to (no
    await in between), so plans are shared between the decoders of one thread and never across threads."""

    def __init__(self):
        self.plans = OrderedDict()


_plan_cache = _PlanCache()


def cached_batch_open(modulus, n, t, z, zc, use_omega_powers=False, degree=None, max_shares=1 << 20, device=None):
    """BatchOpen(...) through the per-thread LRU of plans (HB_PLAN_CACHE entries, default 16; 0 = no caching)."""
    cap = int(os.environ.get("HB_PLAN_CACHE", "16"))
    if cap <= 0:
        return BatchOpen(modulus, n, t, z=z, zc=zc, use_omega_powers=use_omega_powers, degree=degree, max_shares=max_shares, device=device)
    key = (int(modulus), n, t, tuple(z), tuple(zc), bool(use_omega_powers), degree, int(max_shares), device)
    plans = _plan_cache.plans
    plan = plans.pop(key, None)
    if plan is None:
        plan = BatchOpen(modulus, n, t, z=z, zc=zc, use_omega_powers=use_omega_powers, degree=degree, max_shares=max_shares, device=device)
    plans[key] = plan
    while len(plans) > cap:
        plans.popitem(last=False)
    return plan


INT32_MAX = (1 << 31) - 1
_points_cache = {}          # (modulus, n, omega powers?) -> (points as ints, packed host array)
_checked_columns = {}       # id(receive buffer) -> (weak reference to it, the shape / device it was checked for)


class _Probe:
    """hb_probe_*: the reference's per-polynomial Gao decode (reed_solomon.py:151-186) for ONE codeword of the party-major
    buffer, incremental in the arrivals (include/hbmpc_hip.h).  A probe works on the context's side stream: its kernels (one workgroup)
    only read the columns, so feeding it can start -- `feed_ahead` -- while the decoder's own launches and bookkeeping go on."""

    def __init__(self, ctx, xh_all, n, k):
        self.ctx, self.n, self.k = ctx, n, k
        self.h = ctypes.c_void_p()
        rc = ctx.lib.hb_probe_create(ctx.h, np_ptr(xh_all), n, k, ctypes.byref(self.h), ctx.stream())
        if rc != HB_OK:
            self.h = None
            if rc == HB_ERR_UNSUPPORTED:
                raise _Unsupported()
            ctx.check(rc, "hb_probe_create")        # out of memory, bad arguments: real errors, not a reason to change paths
        self.fed, self.poly = [], -1
        self._ok = ctypes.c_int32(0)
        self._mask = np.zeros(n, dtype=np.uint8)
        # (the context's one side stream, not a stream per probe: a process has four hardware queues, and a probe that shares one with the
        # caller's stream feeds behind the decoder's launches instead of beside them)
        self._side_raw = ctypes.c_void_p()
        ctx.check(ctx.lib.hb_side_stream(ctx.h, ctypes.byref(self._side_raw)), "hb_side_stream")

    void_launches = 0

    def _feed(self, z, cols, c, poly, decide, after_current, retried=False):
        if poly != self.poly or z[: len(self.fed)] != self.fed:
            self.ctx.check(self.ctx.lib.hb_probe_reset(self.h), "hb_probe_reset")
            self.fed, self.poly = [], poly
        new = z[len(self.fed):]
        if not new and not decide:
            return
        if after_current:
            # the columns were written on the caller's stream -- copied in by add(idx, column), or received in place and reduced there
            # (ctx.reduce_) before add(idx): the probe's stream always waits for it (one event)
            self.ctx.check(self.ctx.lib.hb_stream_after(self.ctx.h, self._side_raw, self.ctx.stream()), "hb_stream_after")
        ia = np.array(new if new else [0], dtype=np.int32)
        rc = self.ctx.lib.hb_probe_feed(self.h, np_ptr(ia), len(new), self.ctx.ptr(cols), c, poly, 1 if decide else 0, ctypes.byref(self._ok), np_ptr(self._mask),
                                        self._side_raw)
        if rc == HB_ERR_RETRY:
            # a void launch: the workgroups of a probe talk through memory and one gave up waiting for another (a crowded chip).  Nothing
            # was fed and the probe is reset: once more over the whole list on the fewest workgroups the point set allows; a second void
            # launch is the caller's to route around (_ProbeVoid: the batched decoder gives the same verdict)
            self.fed, self.poly = [], -1
            if retried:
                raise _ProbeVoid()
            self.ctx.check(self.ctx.lib.hb_probe_workgroups(self.h, 0), "hb_probe_workgroups")
            self.void_launches += 1
            return self._feed(z, cols, c, poly, decide, False, retried=True)
        self.ctx.check(rc, "hb_probe_feed")
        self.fed = list(z)

    def feed_ahead(self, z, cols, c, poly, after_current=True):
        """enqueue the points of the arrival list z that have not been fed, without asking for a verdict"""
        self._feed(z, cols, c, poly, False, after_current)

    def decide(self, z, cols, c, poly, after_current=True):
        """the verdict over the arrival list z for polynomial `poly`: None, or the sorted list of senders in error"""
        self._feed(z, cols, c, poly, True, after_current)
        if not self._ok.value:
            return None
        return np.nonzero(self._mask)[0].tolist()

    def reset(self):
        """forget everything fed: the next decide() starts a new codeword (a probe changes hands through the pool)"""
        self.ctx.check(self.ctx.lib.hb_probe_reset(self.h), "hb_probe_reset")
        self.fed, self.poly = [], -1

    def close(self):
        if self.h is not None:
            self.ctx.lib.hb_probe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class _CDec:
    """hb_dec_* (include/hbmpc_hip.h): the optimistic phase of one IncrementalDecoder round behind the C ABI -- arrivals are announced by
    index, the object enqueues what depends on the first degree+1 of them, launches decode + validate behind the column that completes
    the quorum and waits for the verdict.  Reusable: begin() starts the next round; pooled per (field, device, points, degree, t)."""

    def __init__(self, ctx, xh_all, n, degree, max_errors):
        self.ctx = ctx
        self.h = ctypes.c_void_p()
        rc = ctx.lib.hb_dec_create(ctx.h, np_ptr(xh_all), n, degree, max_errors, ctypes.byref(self.h), ctx.stream())
        if rc != HB_OK:
            self.h = None
            if rc == HB_ERR_UNSUPPORTED:
                raise _Unsupported()
            ctx.check(rc, "hb_dec_create")
        self.addr = self.h.value
        self._zbuf = np.empty(n, dtype=np.int32)
        self._zptr = np_ptr(self._zbuf)
        self._cnt = ctypes.c_int32(0)
        self._state, self._first = ctypes.c_int32(0), ctypes.c_int32(0)
        self._options = 0

    def settle(self):
        """wait for the verdict of the launch the quorum's arrival enqueued (deferred rounds): -> the state it leads to"""
        return self.ctx.lib.hb_dec_settle(self.h)

    def begin(self, cols, c, n_coef, out, excluded, options=0):
        if options != self._options:
            self.ctx.check(self.ctx.lib.hb_dec_options(self.h, options), "hb_dec_options")
            self._options = options
        ex = np.array(sorted(excluded), dtype=np.int32) if excluded else None
        rc = self.ctx.lib.hb_dec_begin(self.h, self.ctx.ptr(cols), c, n_coef, self.ctx.ptr(out), np_ptr(ex) if ex is not None else None,
                                       len(excluded) if excluded else 0, self.ctx.stream())
        if rc == HB_ERR_UNSUPPORTED:
            return False
        self.ctx.check(rc, "hb_dec_begin")
        return True

    def arrivals(self):
        """the senders counted so far, in arrival order"""
        self.ctx.check(self.ctx.lib.hb_dec_arrivals_list(self.h, self._zptr, len(self._zbuf), ctypes.byref(self._cnt)), "hb_dec_arrivals_list")
        return self._zbuf[: self._cnt.value].tolist()

    def first_bad(self):
        self.ctx.check(self.ctx.lib.hb_dec_verdict(self.h, ctypes.byref(self._state), ctypes.byref(self._first)), "hb_dec_verdict")
        return self._first.value

    def close(self):
        if self.h is not None:
            self.ctx.lib.hb_dec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class _ProbeVoid(Exception):
    """two launches of a probe in a row were void (hb_probe_feed: HB_ERR_RETRY): the caller takes the verdict from the batched decoder"""


class _Unsupported(Exception):
    """the plan-free path does not take this context / shape: the caller uses open plans"""


class _ProbePool(threading.local):
    """idle probes of this thread by (modulus, device, points, k): a decoder borrows one when it first needs it (only in robust
    mode) and hands it back when it is finished"""

    def __init__(self):
        self.idle = {}
        self.quick = {}             # idle _CDec objects by (modulus, device, n, point policy, degree, t); "no" = this point set / context does not qualify


_probe_pool = _ProbePool()


class DeviceIncrementalDecoder:
    """IncrementalDecoder (reference reed_solomon.py:232-403) on device tensors: columns arrive as (C, limbs) tensors
    and stay in one party-major buffer in HBM; the guess, its validation and the robust fallback are launches over all
    C polynomials.  Same state machine and the same decisions as `reed_solomon.IncrementalDecoder`:

      * degree+1 columns: optimistic decode + re-encode (the guess) -- an open plan (BatchOpen) over the arrival set reads
        the arrived rows of the party-major buffer in place and writes the guess party-major: no gather, no transpose;
      * every later column is compared with its row of the guess; degree+1+max_errors-|confirmed| agreeing columns finish;
      * the first disagreement switches to robust mode for good (reference :334-365).  The reference then robust-decodes
        polynomial after polynomial.  Here the next polynomial is robust-decoded alone (Gao or Welch-Berlekamp, one
        codeword); when it shows no error over the current arrival set, ONE plan interpolates every remaining polynomial
        from degree+1 of the arrived columns and checks it against all the other arrived columns: if that passes, every one
        of them would have robust-decoded without errors to exactly these coefficients (the interpolant is unique), and they
        are all accepted at once.  Only a batch in which some later polynomial still hides an error goes through the
        batched robust decode (`_robust_batch`, one launch over the remaining codewords, accepted in order).

    robust: "gao" (what batch_reconstruct uses, batch_reconstruction.py:85-90) or "wb".
    get_results() -> ((C, degree+1, limbs) coefficient tensor, set of confirmed erroneous senders) or (None, None).

    Plan-free fast path (wide contexts the full-size matrix-core kernel takes; `HB_NO_QUICK=1` or an unsupported shape selects
    the plan-based path above, same decisions):
      * optimistic phase: nothing is computed until enough columns are in to finish (degree+1+max_errors-|confirmed|): the guess
        from the first degree+1 arrivals and its comparison with every later arrival are ONE launch then
        (`hb_quick_interp_check`, matrix built on the device).  Before that point the reference's state -- guess, "still
        optimistic" -- has no observable effect: a disagreement only matters once the robust decoder may run, which needs the
        same number of columns.
      * robust phase: the next polynomial's Gao verdict comes from an incremental single-codeword probe (`hb_probe_*`: one small
        kernel per arrival, no n' x n' inverse); once it decodes, its errors are dropped and ONE launch interpolates ALL remaining
        polynomials from degree+1 of the remaining columns and checks them against the rest, reporting the first chunk that
        still disagrees: everything before it is accepted (each would robust-decode to exactly that with no errors), and the
        probe moves on to that chunk.
    """

    # state with an immutable initial value lives on the class until an instance changes it (a decoder is made per open and per round:
    # its constructor is on the path of every open)
    _ch = None                 # address of the hb_dec that runs this round's optimistic phase (None: the Python state machine below)
    _wh = None                 # address of the hb_wait that judges arrivals while candidates wait (None: the memo branch of _fast_robust_update does)
    _wobj = None
    _wbuf = None
    _pending = False           # defer_verdict: the quorum's launch is enqueued, its verdict not read yet (_settle)
    _late = None               # ... and the senders announced since, in order
    _cdec = None
    _fetch1 = None
    _z_epoch = 0               # bumps whenever senders LEAVE the arrival list (between bumps it only grows at the end)
    _optimistic = True
    _guess_decoded = None      # (C, d, limbs)
    _guess_encoded = None      # (n, C, limbs)
    _num_decoded = 0
    _partial_buf = None        # (C, degree+1, limbs): allocated by the robust phase, which alone fills it piecewise
    _result = None
    _last_status = None
    _probe_memo = None         # (polynomial index, arrival list, coefficient ints, error set) of the last successful probe
    _status = None             # (2,) int32 on the device: disagreement flag, first disagreeing chunk
    _probe_obj = None
    _settled = None            # polynomial whose verdict is in (its errors expelled) but which is not accepted yet
    _memo = None               # (polynomial, (arrivals seen, their epoch), candidates [coefficients, who disagrees, values at the n points]) waiting for support
    _prefer_tail = False       # robust phase: interpolate from the newest arrivals (True) or the oldest
    _stalled = None            # polynomial the last robust update could not decode (the probe is on it)
    _probe_next = None         # (polynomial, arrival-list epoch, columns needed before the probe's verdict can matter)
    _checked = None            # (arrival list, first chunk, coefficients, first disagreeing chunk) of a launch the robust phase may reuse
    _scan = None               # robust phase: one launch's coefficients, ALL its disagreeing chunks and who disagrees on each (_Scan)
    radius_verdicts = 0        # polynomials settled by the batched launch's own candidate (diagnostic)
    probes = 0                 # single-codeword robust decodes so far (diagnostic)
    probes_replayed = 0        # probes answered from the previous one (diagnostic)
    launches = 0               # batched robust-decode launches so far (diagnostic)
    plan_accepts = 0           # batches accepted by one interpolate-and-check launch (diagnostic)
    quick_launches = 0         # plan-free interpolate-and-check launches (diagnostic)

    def __init__(self, modulus, n, t, degree=None, batch_size=1, use_omega_powers=False, confirmed_errors=None, device=None, robust="gao",
                 columns=None, want="all", defer_verdict=False, stream_busy=False):
        """columns: an (n, batch_size, limbs) party-major tensor the transport receives into (row j = what party j sent); a column that
        has landed there is announced with add(j) -- no copy.  Without it the decoder keeps a buffer of its own, `slot(j)` is row j of it,
        and add(j, column) copies.
        want: "all" -- get_results() yields every coefficient, (C, degree+1, limbs); "constant" -- only the constant terms are needed
        (what R1 forwards, batch_reconstruction.py:194): a decoder that finishes on its optimistic step then yields (C, 1, limbs).
        defer_verdict: the add() that completes the quorum enqueues decode + validate and returns; `pending()` is True until `done()` or
        `get_results()` (or the next `add`) waits for the verdict.  The reference's coroutine subscribes to both rounds up front
        (batch_reconstruction.py:158-176): with this, the next round's decoder is made -- and its first columns announced -- while this
        round's launch runs.  Same results, same decisions; only where the host waits moves.
        stream_busy: the caller's stream has work enqueued that this round's columns do not wait for (the open's encode; the previous
        round's launch): what depends on the first degree+1 arrivals is built on a stream of the decoder's own, beside it."""
        if robust not in ("gao", "wb"):
            raise ValueError("robust must be 'gao' or 'wb'")
        self.ctx = ctx = Context.get(modulus, device)
        self.n, self.max_errors = n, t
        self.degree = t if degree is None else degree
        self.batch_size = int(batch_size)
        self.robust = robust
        self.use_omega_powers = use_omega_powers
        # the party points and their packed form: per (field, n, policy), not per decoder (a decoder is made per open and per round)
        pk = (int(modulus), n, bool(use_omega_powers))
        pts = _points_cache.get(pk)
        if pts is None:
            from .field import GF
            from .polynomial import EvalPoint

            point = EvalPoint(GF(modulus), n, use_omega_powers=use_omega_powers)
            xs = [point(i).value for i in range(n)]
            pts = _points_cache[pk] = (xs, ctx.host_elems(xs))
        self.x, self._xh_all = pts
        self.L = ctx.n_limbs
        if want not in ("all", "constant"):
            raise ValueError("want must be 'all' or 'constant'")
        self._want_all = want == "all"
        self._in_place = columns is not None
        if columns is None:
            self._cols = ctx.empty(n * self.batch_size).view(n, self.batch_size, self.L)
        else:
            # (a transport hands the same receive buffer to the decoder of every open: its checks are made once per buffer)
            seen = _checked_columns.get(id(columns))
            if seen is not None and seen[0]() is columns and seen[1] == (n, self.batch_size, self.L, ctx.device):
                self._cols = columns
            else:
                if tuple(columns.shape) != (n, self.batch_size, self.L):
                    raise ValueError("columns must be an (n, batch_size, limbs) tensor")
                self._cols = ctx.elems(columns.view(n * self.batch_size, self.L), n * self.batch_size, what="columns").view(n, self.batch_size, self.L)
                if columns.is_contiguous():
                    if len(_checked_columns) >= 16:
                        _checked_columns.clear()
                    _checked_columns[id(columns)] = (weakref.ref(columns), (n, self.batch_size, self.L, ctx.device))
        self._confirmed_errors = set() if confirmed_errors is None else confirmed_errors
        self._avl = set()               # senders counted (reference: _available_points); a property while the C decoder counts them
        self._zl = []                   # ... in arrival order (reference: _z)
        self._defer = bool(defer_verdict)
        self._options = (HB_DEC_OPT_DEFER if defer_verdict else 0) | (HB_DEC_OPT_BESIDE if stream_busy else 0)
        self._fast = ctx.n_limbs == 4 and (self.degree + 1) >= 4 and not os.environ.get("HB_NO_QUICK")   # cleared at the first UNSUPPORTED
        if self._fast:
            self._c_begin()

    # -- the optimistic phase behind the C ABI (hb_dec_*) ---------------------------------------------------------------------------
    def _c_begin(self):
        """hand this round's optimistic phase to a pooled hb_dec: add(idx) is then one C call per arrival (reference reed_solomon.py:367-403
        up to the first verdict); any state change comes back through _c_event"""
        key = (self.ctx.modulus, self.ctx.device, self.n, self.use_omega_powers, self.degree, self.max_errors)
        idle = _probe_pool.quick.get(key)
        if idle is None:
            idle = _probe_pool.quick[key] = []
        if idle == "no":
            return
        if idle:
            cd = idle.pop()
        else:
            try:
                cd = _CDec(self.ctx, self._xh_all, self.n, self.degree, self.max_errors)
            except _Unsupported:
                _probe_pool.quick[key] = "no"
                return
        n_coef = self.degree + 1 if self._want_all else 1
        out = self.ctx.empty(self.batch_size * n_coef)
        excluded = [i for i in self._confirmed_errors if 0 <= i < self.n]
        if not cd.begin(self._cols, self.batch_size, n_coef, out, excluded, self._options):
            idle.append(cd)                  # this shape (nothing to compare, too many coefficients, ...): the Python path; the object serves others
            return
        self._cdec, self._ch, self._cout, self._ckey = cd, cd.addr, out, key

    def _leave_c(self, fetch=True):
        """the C decoder's part is over: its arrival list becomes this object's (fetch), the hb_dec goes back to the pool"""
        cd, self._cdec, self._ch = self._cdec, None, None
        if fetch:
            self._zl = cd.arrivals()
            self._avl = set(self._zl)
        idle = _probe_pool.quick.get(self._ckey)
        if isinstance(idle, list) and len(idle) < 8:
            idle.append(cd)
        else:
            cd.close()
        return cd

    @property
    def _z(self):
        return self._cdec.arrivals() if self._cdec is not None else self._zl

    @_z.setter
    def _z(self, value):
        self._zl = value

    @property
    def _available_points(self):
        return set(self._cdec.arrivals()) if self._cdec is not None else self._avl

    @_available_points.setter
    def _available_points(self, value):
        self._avl = value

    def _c_event(self, state, idx):
        """hb_dec_arrived1 left HB_DEC_COLLECTING at the arrival of `idx` (or failed: state < 0)"""
        if state < 0:
            self.ctx.check(-state, "hb_dec_arrived1")
        if state == HB_DEC_PENDING:
            if self._pending:
                self._late.append(idx)           # announced while the verdict is out: counted once it is in (if the round goes on)
            else:
                self._pending, self._late = True, []
            return
        d = self.degree + 1
        if state == HB_DEC_DONE:
            # (the hb_dec stays with this object until it goes away: its arrival list is only fetched if somebody asks for _z)
            self.quick_launches += 1
            self._ch = None
            self._result = self._cout.view(self.batch_size, d if self._want_all else 1, self.L)
            return
        if state == HB_DEC_DISAGREE:
            self.quick_launches += 1
            first = self._cdec.first_bad()
            self._leave_c()
            self._optimistic = False
            # (a set of confirmed errors shared with other decoders may have grown since this round began: those senders are dropped
            # before the robust phase looks at the list, as the reference drops them at its next add)
            late = [i for i in self._zl if i in self._confirmed_errors]
            if late:
                self._zl = [i for i in self._zl if i not in self._confirmed_errors]
                self._avl = set(self._zl)
                self._z_epoch += 1
            elif self._want_all:
                self._checked = (list(self._zl), 0, self._cout.view(self.batch_size, d, self.L), first)      # the robust phase starts from this very launch
            if len(self._avl) >= self._min_points_required():
                try:
                    self._fast_robust_update()
                    if self._memo is not None:
                        self._w_arm()
                except _Unsupported:
                    self._fast = False
                    self._robust_update()
            return
        if state == HB_DEC_UNSUPPORTED:
            self._leave_c()
            return self._after_arrival(idx)
        raise RuntimeError(f"hb_dec_arrived1: unknown state {state}")

    @property
    def _partial(self):
        if self._partial_buf is None:
            self._partial_buf = self.ctx.empty(self.batch_size * (self.degree + 1)).view(self.batch_size, self.degree + 1, self.L)
        return self._partial_buf

    @_partial.setter
    def _partial(self, value):
        self._partial_buf = value

    def slot(self, idx):
        """row idx of the party-major buffer: where party idx's column is received; add(idx) announces it"""
        return self._cols[idx]

    def accepts(self, idx):
        """would add(idx) count this sender's column?  (not once the decoder is done, nor a sender already counted or confirmed in error:
        reference reed_solomon.py:369-372) -- a transport that receives in place asks BEFORE it writes into slot(idx)"""
        if self._pending and idx in self._late:
            return False
        return self._result is None and idx not in self._confirmed_errors and idx not in self._available_points

    # -- kernels ---------------------------------------------------------------------------------
    def _plan(self, z, zc):
        return cached_batch_open(self.ctx.modulus, self.n, self.max_errors, z, zc, use_omega_powers=self.use_omega_powers,
                                 degree=self.degree, max_shares=self.batch_size * (self.degree + 1), device=self.ctx.device)

    def _interpolate_and_check(self, z, zc):
        """coefficients of every polynomial from the arrived rows z of the party-major buffer, validated against rows zc:
        -> ((C, d, limbs), all agreed?)"""
        d = self.degree + 1
        plan = self._plan(z, zc)
        dec = plan.r2_decode(self._cols.view(self.n * self.batch_size, self.L), self.batch_size * d)
        ok = plan.ok() if zc else True
        return dec.view(self.batch_size, d, self.L), ok, plan

    def _rows(self, lo, count=None, order=None):
        """the arrived columns of polynomials lo.. as (count, npts, limbs), one codeword per row (robust decodes only)"""
        idx = self.ctx.torch.tensor(self._z if order is None else order, dtype=self.ctx.torch.int64, device=self.ctx.tdev)
        hi = self.batch_size if count is None else min(self.batch_size, lo + count)
        return self._cols[:, lo:hi, :].index_select(0, idx).transpose(0, 1).contiguous()

    def _decode_and_encode(self):
        c, d, n = self.batch_size, self.degree + 1, self.n
        dec, _, plan = self._interpolate_and_check(list(self._z), [])
        enc = plan.r1_encode(dec.view(c * d, self.L))                 # the guess as a share vector: chunk c = its coefficients
        self._guess_decoded = dec
        self._guess_encoded = enc.view(n, c, self.L)

    def _robust_batch(self, limit=None):
        """robust decode of the remaining polynomials (the first `limit` of them) over the current arrival set:
        -> ok (Crem,) bool, coeffs (Crem, d, limbs), errs (Crem, n) bool: the senders in error."""
        ctx, t = self.ctx, self.ctx.torch
        lo, d, n, npts = self._num_decoded, self.degree + 1, self.n, len(self._z)
        crem = self.batch_size - lo if limit is None else min(limit, self.batch_size - lo)
        co = ctx.empty(crem * d)
        if limit is None:
            self.launches += 1
        else:
            self.probes += 1
        if self.robust == "wb":
            # reed_solomon.py:203-224: decode over the arrived points (in party order, as the reference enumerates them),
            # re-encode, the disagreeing positions are the errors
            zs = sorted(self._z)
            rows = self._rows(lo, crem, zs)
            xz = ctx.host_elems([self.x[i] for i in zs])
            present = t.ones(crem * npts, dtype=t.uint8, device=ctx.tdev)
            ln = t.zeros(crem, dtype=t.int32, device=ctx.tdev)
            st = t.zeros(crem, dtype=t.int32, device=ctx.tdev)
            ctx.check(ctx.lib.hb_wb_decode(ctx.h, np_ptr(xz), npts, d, ctx.ptr(rows), ctx.ptr(present), crem, ctx.ptr(co), ctx.ptr(ln), ctx.ptr(st), ctx.stream()), "wb")
            ok = st == 0
            self._last_status = st
            ev = ctx.empty(crem * npts)
            ctx.check(ctx.lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(xz), npts, ctx.ptr(co), crem, d, ctx.ptr(ev), ctx.stream()), "evaluate")
            differs = (ev.view(crem, npts, self.L) != rows).any(dim=2) & ok.unsqueeze(1)
            errs = t.zeros((crem, n), dtype=t.bool, device=ctx.tdev)
            errs[:, t.tensor(zs, dtype=t.int64, device=ctx.tdev)] = differs
            return ok, co.view(crem, d, self.L), errs
        rows = self._rows(lo, crem)
        xz = ctx.host_elems([self.x[i] for i in self._z])
        self._last_status = None
        el = t.zeros((crem * (npts + 1), self.L), dtype=t.int64, device=ctx.tdev)
        ln = t.zeros(crem, dtype=t.int32, device=ctx.tdev)
        ok = t.zeros(crem, dtype=t.uint8, device=ctx.tdev)
        ctx.check(ctx.lib.hb_gao_decode(ctx.h, np_ptr(xz), npts, d, ctx.ptr(rows), crem, ctx.ptr(co), ctx.ptr(el), ctx.ptr(ln), ctx.ptr(ok), ctx.stream()), "gao")
        ok = ok.bool()
        # roots of the error locator among ALL party points are the faulty senders (reference :174-184); a locator of
        # length <= 1 names nobody.  Entries past the locator's length are not part of it.
        keep = t.arange(npts + 1, device=ctx.tdev).unsqueeze(0) < ln.unsqueeze(1)
        el = (el.view(crem, npts + 1, self.L) * keep.unsqueeze(2)).contiguous()
        ev = ctx.empty(crem * n)
        ctx.check(ctx.lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(self._xh_all), n, ctx.ptr(el), crem, npts + 1, ctx.ptr(ev), ctx.stream()), "evaluate")
        errs = (ev.view(crem, n, self.L) == 0).all(dim=2) & (ln > 1).unsqueeze(1) & ok.unsqueeze(1)
        return ok, co.view(crem, d, self.L), errs

    def _undecodable(self, pos):
        """polynomial `pos` of the last robust batch did not decode: (None, None) for the reference's swallowed failures,
        its other exceptions re-raised as the reference's robust decoder does (reed_solomon.py:205-212)"""
        if self._last_status is not None:
            st = int(self._last_status[pos].item())
            if st == 2:
                raise Exception("No solution")           # reed_solomon_wb.py:245
            if st == 3:
                raise AssertionError("2 * t + 1 + c <= n")   # reed_solomon_wb.py:132

    # -- plan-free kernels (hb_quick.hip) ----------------------------------------------------------------
    def _quick(self, z, zc, store=True, lo=0, hi=None, want_map=False):
        """interpolate every polynomial from chunk `lo` on from the arrived rows z, compare with the arrived rows zc, in one launch:
        -> ((C, d, limbs) | None, all agreed?, first disagreeing chunk); raises _Unsupported when the kernel does not take it.
        want_map: a fourth value, the list of ALL disagreeing chunks (None when they all agreed)"""
        ctx, t = self.ctx, self.ctx.torch
        d = self.degree + 1
        if self._status is None:
            self._status = self._status_init().clone()
        out = ctx.empty(self.batch_size * d) if store else None
        za = np.array(z, dtype=np.int32)
        zca = np.array(zc if zc else [0], dtype=np.int32)
        hi_ = self.batch_size if hi is None else hi
        bad_map = t.zeros((hi_ - lo + 31) // 32 + 1, dtype=t.int32, device=ctx.tdev) if want_map and zc else None
        rc = ctx.lib.hb_quick_interp_check_map(ctx.h, np_ptr(self._xh_all), self.n, np_ptr(za), d, np_ptr(zca), len(zc),
                                               ctx.ptr(self._cols), self.batch_size, lo, hi_,
                                               ctx.ptr(out) if store else None, ctx.ptr(self._status),
                                               ctx.ptr(bad_map) if bad_map is not None else None, ctx.stream())
        if rc == HB_ERR_UNSUPPORTED:
            raise _Unsupported()
        ctx.check(rc, "hb_quick_interp_check_map")
        self.quick_launches += 1
        dec = out.view(self.batch_size, d, self.L) if store else None
        if not zc:
            return (dec, True, INT32_MAX, None) if want_map else (dec, True, INT32_MAX)
        flag, first = self._status.tolist()          # synchronises
        if flag:
            self._status.copy_(self._status_init())
        res = (dec, not flag, (first + lo if flag else INT32_MAX))
        if not want_map:
            return res
        bad = None
        if flag and bad_map is not None:
            bits = np.unpackbits(bad_map.cpu().numpy().view(np.uint8), bitorder="little")
            bad = (np.nonzero(bits)[0] + lo).tolist()
        return res + (bad,)

    def _status_init(self):
        init = getattr(self.ctx, "_quick_status_init", None)
        if init is None:
            init = self.ctx._quick_status_init = self.ctx.torch.tensor([0, INT32_MAX], dtype=self.ctx.torch.int32, device=self.ctx.tdev)
        return init

    def _borrow_probe(self):
        if self._probe_obj is None:
            key = (self.ctx.modulus, self.ctx.device, self.n, self.degree + 1, self.use_omega_powers)
            idle = _probe_pool.idle.setdefault(key, [])
            if idle:
                # a pooled probe still holds its last owner's points (host list, C-side list and device state): same polynomial index
                # and an arrival list that extends the old one would otherwise skip decide()'s reset and judge another open's data
                self._probe_obj = idle.pop()
                self._probe_obj.reset()
            else:
                self._probe_obj = _Probe(self.ctx, self._xh_all, self.n, self.degree + 1)
            self._probe_key = key
        return self._probe_obj

    def _return_probe(self):
        pr, self._probe_obj = self._probe_obj, None
        if pr is not None and pr.h is not None:
            idle = _probe_pool.idle.setdefault(self._probe_key, [])
            if len(idle) < 8:
                pr.reset()
                idle.append(pr)
            else:
                pr.close()

    def __del__(self):
        try:
            self._return_probe()
            if self._wobj is not None:
                self.ctx.lib.hb_wait_destroy(self._wobj)
                self._wobj = None
            if self._cdec is not None:
                if self._pending:
                    self._cdec.settle()          # (the launch writes this round's result tensor: it must have ended before the tensor goes)
                self._leave_c(fetch=False)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _fast_optimistic(self):
        """enough columns to finish: the guess from the first degree+1 arrivals against every later one.  True = done.  (Rounds the C
        decoder does not take: hb_dec_begin answered HB_ERR_UNSUPPORTED.)"""
        d = self.degree + 1
        dec, agree, first = self._quick(self._z[:d], self._z[d:])
        if agree:
            self._result = dec
            return True
        self._optimistic = False
        if dec is not None:
            self._checked = (list(self._z), 0, dec, first)      # the robust phase starts from this very launch
        return False

    def _probe_ahead(self, poly):
        """polynomial `poly` disagrees: its Gao verdict may be needed in a moment (when no candidate of the batched launch lies within the
        radius).  The probe catches up with the arrival list on its own stream meanwhile -- one workgroup, it only reads the columns."""
        if self.robust != "gao" or not self._fast or poly >= self.batch_size:
            return
        try:
            try:
                self._borrow_probe().feed_ahead(self._z, self._cols, self.batch_size, poly)
            except _ProbeVoid:
                pass                                         # (nothing was fed: the next verdict feeds the whole list)
        except _Unsupported:
            pass

    def _disagreeing(self, coeffs):
        """the arrived senders whose symbol of ONE polynomial differs from `coeffs` ((d, limbs)) evaluated at their point"""
        ctx, t = self.ctx, self.ctx.torch
        d = self.degree + 1
        ev = ctx.empty(self.n)
        ctx.check(ctx.lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(self._xh_all), self.n, ctx.ptr(coeffs.contiguous()), 1, d, ctx.ptr(ev), ctx.stream()), "evaluate")
        return ev

    def _split(self, tail):
        """degree+1 arrived columns to interpolate from and the rest to compare with: the oldest or the newest arrivals"""
        d = self.degree + 1
        return (self._z[-d:], self._z[:-d]) if tail else (self._z[:d], self._z[d:])

    def _candidate_cap(self, radius):
        """How many disagreeing senders a candidate may have and still decide the reference's verdicts (reed_solomon.py:334-346).
        The reference accepts a robust decode (Q, errors) only when |z| - |errors| >= need = degree + 1 + max_errors - confirmed
        (:343-345) and otherwise waits with nothing changed.  Let P be ANY polynomial of degree <= `degree` that disagrees with E of
        the arrived senders, E <= need - (degree + 1) = max_errors - confirmed.  An accepted Q agrees with >= need of the |z| columns and P
        with |z| - E of them, so they agree with each other on >= need - E >= degree + 1 points: Q = P.  And once |z| - E >= need, E <=
        (|z| - degree - 1) / 2: P is inside Gao's unique-decoding radius, so Gao returns exactly P with exactly those E senders.  Hence
        while E stays within the cap, "accept P when |z| - E >= need, else wait" IS the reference's behaviour -- whatever Gao would have said
        in between (None, or some other polynomial short of support) changes nothing.  A candidate from ANY degree + 1 columns serves;
        with the liars among the first arrivals the newest columns give the true polynomial and no incremental decode runs at all.
        Welch-Berlekamp answers (or raises) in its own way beyond the radius: for it the cap is the radius."""
        return max(radius, self.max_errors - len(self._confirmed_errors)) if self.robust == "gao" else radius

    def _symbols(self, chunk, senders):
        """the symbols of `chunk` in the columns of `senders`, on the host: (len(senders), limbs)"""
        ctx = self.ctx
        if len(senders) == 1:
            # one arrival, the case that repeats: buffers and their addresses are kept
            fx = self._fetch1
            if fx is None:
                ia, out = np.empty(1, dtype=np.int32), np.empty((1, self.L), dtype=np.int64)
                fx = self._fetch1 = (ia, out, np_ptr(ia), np_ptr(out), ctx.ptr(self._cols))
            fx[0][0] = senders[0]
            ctx.check(ctx.lib.hb_symbols_fetch(ctx.h, fx[4], self.n, self.batch_size, chunk, fx[2], 1, fx[3], ctx.stream()), "hb_symbols_fetch")
            return fx[1]
        out = np.empty((len(senders), self.L), dtype=np.int64)
        for lo in range(0, len(senders), 64):
            part = np.asarray(senders[lo:lo + 64], dtype=np.int32)
            ctx.check(ctx.lib.hb_symbols_fetch(ctx.h, ctx.ptr(self._cols), self.n, self.batch_size, chunk, np_ptr(part), len(part), np_ptr(out[lo:lo + 64]), ctx.stream()),
                      "hb_symbols_fetch")
        return out

    def _track_candidates(self, cands, chunk, senders):
        """each candidate [coeffs, errors, values at the n points (host, made on first use)] with `senders` (new arrivals) judged; those
        whose disagreements left the cap are dropped"""
        if senders:
            sym = self._symbols(chunk, senders)
            for cand in cands:
                if cand[2] is None:
                    cand[2] = self._disagreeing(cand[0]).cpu().numpy()
                ev = cand[2]
                cand[1] = cand[1] + [s_ for j, s_ in enumerate(senders) if (sym[j] != ev[s_]).any()]
        cap = self._candidate_cap((len(self._z) - self.degree - 1) // 2)
        return [cand for cand in cands if len(cand[1]) <= cap]

    def _candidate_errors(self, coeffs, chunk):
        """-> (the arrived senders whose symbol of `chunk` differs from the candidate at their point, the candidate's values at the n points
        on the host): one small evaluation, its values and the arrived symbols of the chunk brought over, compared there"""
        ctx = self.ctx
        if self.n <= 1024:
            # one launch: the candidate at every party's point and, per party, whether its symbol of the chunk differs -- through pinned memory
            ev = np.empty((self.n, self.L), dtype=np.int64)
            diff = np.empty(self.n, dtype=np.uint8)
            ctx.check(ctx.lib.hb_candidate_check(ctx.h, np_ptr(self._xh_all), self.n, ctx.ptr(coeffs.contiguous()), self.degree + 1, ctx.ptr(self._cols),
                                                 self.batch_size, chunk, np_ptr(ev), np_ptr(diff), ctx.stream()), "hb_candidate_check")
            return [s_ for s_ in self._z if diff[s_]], ev
        ev = self._disagreeing(coeffs).cpu().numpy()
        sym = self._symbols(chunk, self._z)
        differs = (sym != ev[np.asarray(self._z, dtype=np.int64)]).any(axis=1)
        return [s_ for s_, df in zip(self._z, differs) if df], ev

    # One launch names EVERY chunk some compared sender disagrees on (hb_quick_interp_check_map), and the candidates of an interpolation
    # set do not change when a compared sender is expelled: while the interpolation set stands and no new column has arrived, the
    # next disagreeing chunk and the senders that disagree on it are read off a table instead of another launch.  Liars that corrupt
    # one late chunk each cost a launch and a table, not a launch and an evaluation apiece.
    _SCAN_CAP = 128              # chunks a table is built for (a sender that corrupts everything is found at its first chunk anyway)

    def _scan_or_quick(self, tail_split, lo):
        zi, zcmp = self._split(tail_split)
        sc = self._scan
        if sc is not None and sc["interp"] == tuple(zi) and sc["lo"] <= lo and set(zcmp) <= sc["check"]:
            live = set(zcmp)
            for b in sc["bad"]:
                if b >= lo and not live.isdisjoint(sc["table"][b]):
                    return sc["dec"], False, b
            return sc["dec"], True, INT32_MAX
        self._scan = None
        dec, agree, first, bad = self._quick(zi, zcmp, lo=lo, want_map=True)
        if not agree and bad and len(bad) <= self._SCAN_CAP:
            self._scan = {"interp": tuple(zi), "check": set(zcmp), "lo": lo, "dec": dec, "bad": bad, "table": self._disagreement_table(dec, bad)}
        return dec, agree, first

    def _disagreement_table(self, dec, chunks):
        """chunk -> the arrived senders whose symbol of that chunk differs from the candidate dec[chunk] evaluated at their point"""
        ctx, t = self.ctx, self.ctx.torch
        d = self.degree + 1
        k = len(chunks)
        idx = t.tensor(chunks, dtype=t.int64, device=ctx.tdev)
        coeffs = dec.index_select(0, idx).contiguous()                       # (k, d, limbs)
        ev = ctx.empty(self.n * k)
        ctx.check(ctx.lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(self._xh_all), self.n, ctx.ptr(coeffs), k, d, ctx.ptr(ev), ctx.stream()), "evaluate")
        differs = (ev.view(k, self.n, self.L).transpose(0, 1) != self._cols.index_select(1, idx)).any(dim=2).cpu().numpy()      # (n, k)
        arrived = set(self._z)
        return {c: {s for s in np.nonzero(differs[:, j])[0].tolist() if s in arrived} for j, c in enumerate(chunks)}

    def _scan_errors(self, dec, first):
        """-> (errors, the candidate's values at the n points or None when the table answered)"""
        sc = self._scan
        if sc is not None and dec is sc["dec"] and first in sc["table"]:
            return [s for s in self._z if s in sc["table"][first]], None
        return self._candidate_errors(dec[first], first)

    def _wb_refusal(self, lo):
        """Welch-Berlekamp only: raise what the reference's decoder raises when it is asked to decode over fewer than
        2 degree + 1 columns.  Polynomial lo is exempt while its verdict is in (`_settled`): it was decoded over the longer list."""
        if self.robust == "wb" and len(self._z) < 2 * self.degree + 1 and (self._settled != lo or lo + 1 < self.batch_size):
            raise AssertionError("2 * t + 1 + c <= n")

    def _expel(self, errors):
        es = set(errors)
        self._confirmed_errors |= es
        self._available_points -= es
        self._z = [i for i in self._z if i not in es]
        self._z_epoch += 1

    # -- the candidates' waiting phase behind the C ABI (hb_wait_*) ------------------------------------------------------------------
    def _w_arm(self):
        """candidates are parked (self._memo) and every arrival from now on is one question -- does the new sender's symbol of that chunk
        equal the candidates' values at its point?  While the columns are received in place that question is asked in C (hb_wait_arrived1
        behind add(), csrc/hb_pymarshal.c) and this object hears of an arrival only when a candidate can be accepted or none is left
        (_w_event).  Same decisions as the memo branch of _fast_robust_update, which still serves copied columns and Welch-Berlekamp."""
        if self._wh is not None or self.robust != "gao" or not self._in_place or self._result is not None or _marshal is None or not hasattr(_marshal, "bind_wait"):
            return
        lo, (seen, epoch), cands = self._memo
        if seen != len(self._zl) or epoch != self._z_epoch or not cands or len(cands) > 8 or self._cdec is not None:
            return
        for cand in cands:
            if cand[2] is None:
                cand[2] = self._disagreeing(cand[0]).cpu().numpy()
        ctx = self.ctx
        if self._wobj is None:
            h = ctypes.c_void_p()
            ctx.check(ctx.lib.hb_wait_create(ctx.h, self.n, ctypes.byref(h)), "hb_wait_create")
            self._wobj = h
        wb = self._wbuf
        if wb is None:
            ev, counts = np.empty((8, self.n, self.L), dtype=np.int64), np.empty(8, dtype=np.int32)
            wb = self._wbuf = (ev, counts, np_ptr(ev), np_ptr(counts))
        for i, cand in enumerate(cands):
            wb[0][i] = cand[2]
            wb[1][i] = len(cand[1])
        ctx.check(ctx.lib.hb_wait_begin(self._wobj, ctx.ptr(self._cols), self.batch_size, lo, self.degree, self.max_errors, len(self._zl), len(cands),
                                        wb[2], wb[3], ctx.stream()), "hb_wait_begin")
        self._wh = self._wobj.value

    def _w_disarm(self):
        """the wait is over (or abandoned): the contradictions hb_wait counted go to the parked candidates, those that left the cap are dropped"""
        self._wh = None
        lo, _, cands = self._memo
        ctx, alive = self.ctx, []
        buf = np.empty(self.n, dtype=np.int32)
        standing, cnt = ctypes.c_int32(0), ctypes.c_int32(0)
        for i, cand in enumerate(cands):
            ctx.check(ctx.lib.hb_wait_result(self._wobj, i, ctypes.byref(standing), np_ptr(buf), self.n, ctypes.byref(cnt)), "hb_wait_result")
            if standing.value:
                cand[1] = cand[1] + buf[: cnt.value].tolist()
                alive.append(cand)
        self._memo = (lo, (len(self._zl), self._z_epoch), alive)

    def _w_event(self, state, idx):
        """hb_wait_arrived1 ended the wait at the arrival of `idx` (already in the arrival list), or failed (state < 0)"""
        if state < 0:
            self._wh = None
            self._memo = None                      # (what the object counted is void: the robust phase starts from a fresh launch)
            self.ctx.check(-state, "hb_wait_arrived1")
        self._w_disarm()
        return self._after_arrival(idx)

    def _fast_robust_update(self):
        """reference :334-365, plan-free (see the class docstring); robust decoder Gao or Welch-Berlekamp.

        One launch interpolates every open polynomial from degree+1 of the arrived columns and compares with the rest.  Everything
        before the first disagreeing chunk m is accepted (each of those polynomials robust-decodes to exactly that, with no error).
        For polynomial m the launch has also produced a candidate: if the senders that disagree with it number at most
        floor((|z| - degree - 1) / 2), the candidate is within the unique-decoding radius of the received word, hence IS what Gao
        returns, and those senders are its errors -- no decode needed.  Otherwise the interpolation set itself was contaminated
        and the incremental probe (hb_probe_*) gives Gao's verdict; while it says "undecodable", later arrivals only cost its
        update."""
        t = self.ctx.torch
        d = self.degree + 1
        while self._num_decoded < self.batch_size:
            lo = self._num_decoded
            # The reference's Welch-Berlekamp decoder refuses before it looks at the data when fewer than 2 degree + 1 columns are
            # left (reed_solomon_wb.py:132).  Polynomial lo may already have its verdict (errors expelled, not accepted yet): it
            # was decoded over the longer list; every polynomial after it meets the refusal.
            self._wb_refusal(lo)
            if self._stalled == lo:
                if self.robust == "gao":
                    # A verdict that cannot change anything is not asked for.  If Gao failed over m columns, every polynomial is contradicted by
                    # more than r = (m - degree - 1) // 2 of them, and by no fewer as more arrive; if it decoded with e errors, that polynomial
                    # is the only one within r >= e, so every polynomial has at least e: either way the reference, which accepts only when
                    # |z| - |errors| >= need (reed_solomon.py:343-345) and otherwise waits with nothing changed, waits until need + (that
                    # bound) columns are in.  The columns that arrive meanwhile are fed to the probe in one launch at the next verdict
                    # (85 liars spread over the arrival list at n = 256: 8 verdicts instead of 85).
                    pn = self._probe_next
                    if pn is not None and pn[0] == lo and pn[1] == self._z_epoch and len(self._z) < pn[2]:
                        return
                    pr = self._borrow_probe()
                    self.probes += 1
                    try:
                        errors = pr.decide(self._z, self._cols, self.batch_size, lo)
                    except _ProbeVoid:
                        # never out of add(): Gao's verdict for this one polynomial from the batched kernels (same decision, ~0.7 ms)
                        ok, _, errs = self._robust_batch(1)
                        errors = t.nonzero(errs[0]).flatten().tolist() if bool(ok[0].item()) else None
                    if errors is None:
                        self._probe_next = (lo, self._z_epoch, self._min_points_required() + (len(self._z) - d) // 2 + 1)
                        return                               # (None, None): more columns needed
                    self._probe_next = (lo, self._z_epoch, self._min_points_required() + len(errors))
                else:
                    # Welch-Berlekamp beyond the unique-decoding radius answers in its own way (or raises): always the real decode
                    ok, _, errs = self._robust_batch(1)
                    if not bool(ok[0].item()):
                        self._undecodable(0)                 # raises what the reference re-raises
                        return
                    errors = t.nonzero(errs[0]).flatten().tolist()
                if len(self._available_points) - len(errors) < self._min_points_required():
                    return
                self._stalled = None
                self._expel(errors)
                self._settled = lo
            memo, self._memo = self._memo, None
            if memo is not None and memo[0] == lo and memo[1][1] == self._z_epoch:
                # Candidates found earlier for this polynomial, each with the senders that disagree with it (see _candidate_cap for
                # why such a candidate decides Gao's verdicts while its disagreements stay within max_errors - confirmed): only the
                # columns that arrived since are compared, one 32-byte symbol each.
                alive, accepted = self._track_candidates(memo[2], lo, self._z[memo[1][0]:]), None
                for cand in alive:
                    if len(self._available_points) - len(cand[1]) >= self._min_points_required():
                        accepted = cand
                        break
                if accepted is not None:
                    self.radius_verdicts += 1
                    self._expel(accepted[1])
                    self._settled = lo
                elif alive:
                    self._memo = (lo, (len(self._z), self._z_epoch), alive)
                    return
                elif self.robust == "gao" and self._probe_obj is not None and self._probe_obj.poly == lo:
                    # every candidate fell and the probe has been following this polynomial in the background (below): its verdict is the
                    # reference's, two more batched launches for fresh candidates are not worth their 0.2-1.1 ms
                    self._stalled = lo
                    continue
            self._wb_refusal(lo)                             # the list may just have shrunk (expulsions above)
            chk, self._checked = self._checked, None
            tail_split = self._prefer_tail
            if chk is not None and chk[0] == self._z and chk[1] == lo:
                dec, first = chk[2], chk[3]
                agree, tail_split = False, False
            else:
                dec, agree, first = self._scan_or_quick(tail_split, lo)
            if agree:
                if lo == 0:
                    self._partial = dec              # nothing accepted before: the launch's output is the result
                else:
                    self._partial[lo:] = dec[lo:]
                self._num_decoded = self.batch_size
                self.plan_accepts += 1
                break
            if first > lo:
                self._partial[lo:first] = dec[lo:first]
                self._num_decoded = first
            # polynomial `first`: is the candidate the launch produced within the radius?  Any degree+1 of the arrived columns
            # interpolate a candidate; a sender that lies in this chunk may sit among them, so the other end of the arrival
            # list gets one try too (whichever end worked is tried first from then on).
            radius = (len(self._z) - d) // 2
            cap = self._candidate_cap(radius)
            cands = []
            errors, ev = self._scan_errors(dec, first)
            if len(errors) <= cap:
                cands.append([dec[first].clone() if ev is None else None, list(errors), ev])   # (the coefficients only serve to make ev)
            if len(errors) > radius and len(self._z) > d:
                if len(errors) > cap:
                    self._probe_ahead(first)             # (a second candidate is tried first; the probe catches up meanwhile)
                tail_split = not tail_split
                dec2, _, _ = self._quick(*self._split(tail_split), lo=first, hi=first + 1)      # this one polynomial only
                errors, ev = self._candidate_errors(dec2[first], first)
                if len(errors) <= cap:
                    cands.insert(0 if len(errors) <= radius else len(cands), [None, list(errors), ev])
            if len(errors) <= radius:
                self._prefer_tail = tail_split
                if len(self._available_points) - len(errors) < self._min_points_required():
                    self._memo = (first, (len(self._z), self._z_epoch), cands)
                    return
                self.radius_verdicts += 1
                self._expel(errors)
                self._settled = first
                continue
            if cands:
                # outside the radius today, but Gao can accept nothing else while these stand (_candidate_cap): no probe verdict, wait.  The
                # wait lasts several arrivals (|z| - E >= need is far), so the probe catches up on its own stream meanwhile: should the
                # candidates fall -- liars that do not arrive first -- its verdict is at hand without the 43 / 171-point feed on the path
                self._probe_ahead(first)
                self._memo = (first, (len(self._z), self._z_epoch), cands)
                return
            self._stalled = first                        # only the probe can say when it becomes decodable
        if self._num_decoded == self.batch_size:
            self._result = self._partial
            self._scan = None                                # (its launch's coefficient buffer goes with it)
            self._return_probe()

    # -- the state machine (reference :288-372) ------------------------------------------------------
    def _min_points_required(self):
        return self.degree + 1 + self.max_errors - len(self._confirmed_errors)

    def _optimistic_update(self, idx):
        agree = True
        if len(self._available_points) == self.degree + 1:
            self._decode_and_encode()
        else:
            agree = bool(self.ctx.torch.equal(self._cols[idx], self._guess_encoded[idx]))
            if not agree:
                self._guess_decoded = self._guess_encoded = None
                self._optimistic = False
        if agree and len(self._available_points) >= self._min_points_required():
            self._result = self._guess_decoded
        return agree

    def _accept(self, coeffs, errors):
        self._partial[self._num_decoded] = coeffs
        self._num_decoded += 1
        if errors:
            self._confirmed_errors |= set(errors)
            self._available_points -= set(errors)
            self._z = [i for i in self._z if i not in errors]
            self._z_epoch += 1

    def _probe(self):
        """The reference's next robust_decode (one polynomial over the current arrival set): -> (coeffs (d, limbs) | None, errors).
        Gao's answer is determined by its inputs: the unique polynomial of degree < d within floor((|z| - d) / 2) errors of the
        points, and the senders where it disagrees.  When the previous probe of the SAME polynomial succeeded over a prefix of
        the current arrival list, that polynomial evaluated at the new points says whether each of them agrees; as long as the
        disagreeing senders stay within the new radius, the decode over the longer list is the same polynomial with the
        enlarged error set -- no launch needed.  (Welch-Berlekamp beyond the radius may answer differently: it is always run.)"""
        t = self.ctx.torch
        d, lo = self.degree + 1, self._num_decoded
        memo = self._probe_memo
        if self.robust == "gao" and memo is not None and memo[0] == lo and self._z[: len(memo[1])] == memo[1] and len(self._z) > len(memo[1]):
            _, z_old, ints, errors, coeffs = memo
            errors = set(errors)
            p = self.ctx.modulus
            for idx in self._z[len(z_old):]:
                want = 0
                for c in reversed(ints):
                    want = (want * self.x[idx] + c) % p
                got = self.ctx.download_ints(self._cols[idx, lo : lo + 1])[0]
                if got != want:
                    errors.add(idx)
            if len(errors) <= (len(self._z) - d) // 2:
                self._probe_memo = (lo, list(self._z), ints, set(errors), coeffs)
                self.probes_replayed += 1
                return coeffs, sorted(errors)
        ok, coeffs, errs = self._robust_batch(1)
        if not bool(ok[0].item()):
            self._undecodable(0)
            self._probe_memo = None
            return None, None
        errors = t.nonzero(errs[0]).flatten().tolist()
        if self.robust == "gao":
            self._probe_memo = (lo, list(self._z), self.ctx.download_ints(coeffs[0]), set(errors), coeffs[0].clone())
        return coeffs[0], errors

    def _robust_update(self):
        t = self.ctx.torch
        d = self.degree + 1
        while self._num_decoded < self.batch_size:
            # the reference's next robust_decode, alone: while it stalls (undecodable, or too few points once its errors are
            # dropped) nothing else can be accepted either, and an arrival costs one codeword instead of all of them
            coeffs0, errors = self._probe()
            if coeffs0 is None:
                return
            if len(self._available_points) - len(errors) < self._min_points_required():
                return
            if errors:
                self._accept(coeffs0, errors)            # this polynomial confirmed senders in error: the next one sees fewer points
                continue
            # no error in this polynomial over the arrival set.  If every remaining polynomial interpolates from d of the
            # arrived columns and agrees with all the others, each of them robust-decodes to that interpolant with no
            # errors and passes the same agreement count as this one: accept them all.
            dec, all_agree, _ = self._interpolate_and_check(self._z[:d], self._z[d:])
            if all_agree:
                self._partial[self._num_decoded :] = dec[self._num_decoded :]
                self._num_decoded = self.batch_size
                self.plan_accepts += 1
                break
            # some later polynomial still has an error: batched robust decode, accepted in order up to the first such one
            ok, coeffs, errs = self._robust_batch()
            has_err = errs.any(dim=1)
            stop = (~ok) | has_err
            first = int(t.nonzero(stop)[0].item()) if bool(stop.any().item()) else int(ok.shape[0])
            if first:                                    # error-free polynomials before it: accepted as they are
                self._partial[self._num_decoded : self._num_decoded + first] = coeffs[:first]
                self._num_decoded += first
            if first == ok.shape[0]:
                break
            if not bool(ok[first].item()):
                self._undecodable(first)
                return                                   # (None, None): more columns needed
            errors = t.nonzero(errs[first]).flatten().tolist()
            if len(self._available_points) - len(errors) < self._min_points_required():
                return
            self._accept(coeffs[first], errors)
        if self._num_decoded == self.batch_size:
            self._result = self._partial

    def _add_slow(self, idx, column=None):
        """column: (C, limbs) limb tensor on the device, or a list of C ints; None: the column has been received into slot(idx)
        (row idx of the `columns` buffer).  (`add` itself is the C binding below: an in-place arrival while the C decoder runs the
        optimistic phase never gets here.)"""
        if self._result is not None or idx in self._confirmed_errors:
            return
        cd = self._cdec
        if idx in (self._available_points if cd is None else cd.arrivals()):
            return
        if self._wh is not None:
            self._w_disarm()                         # (a column that is copied in, keyword arguments: this arrival is the memo branch's)
        if column is not None:
            if not hasattr(column, "shape"):
                if len(column) != self.batch_size:
                    raise ValueError("Incorrect length of data")
                column = self.ctx.upload_ints(column)
            if tuple(column.shape) != (self.batch_size, self.L):
                raise ValueError("Incorrect length of data")
            column = self.ctx.elems(column, self.batch_size, what="column")
            self._cols[idx] = column
        if cd is not None:
            st = self.ctx.lib.hb_dec_arrived1(cd.h, idx)
            if st:
                self._c_event(st, idx)
            return
        self._avl.add(idx)
        self._zl.append(idx)
        return self._after_arrival(idx)

    def _after_arrival(self, idx):
        """`idx` has just been appended to the arrival list (reference :374-403)"""
        k = len(self._zl)
        if k <= self.degree:
            return
        if self._fast and self._optimistic:
            # nothing is computed before enough columns are in to finish (until then the reference's guess has been compared with nothing)
            if k < self.degree + 1 + self.max_errors - len(self._confirmed_errors):
                return
        if self._fast:
            try:
                return self._fast_add()
            except _Unsupported:
                self._fast = False                       # this context / shape: open plans from here on (same decisions)
                if self._optimistic and self._guess_decoded is None and len(self._available_points) > self.degree + 1:
                    self._catch_up_optimistic()
                    if self.done() or self._optimistic:
                        return
                    if len(self._available_points) >= self._min_points_required():
                        self._robust_update()
                    return
        if self._optimistic and self._optimistic_update(idx):
            return
        if len(self._available_points) >= self._min_points_required():
            self._robust_update()

    # add(idx, column=None) -- reference IncrementalDecoder.add (reed_solomon.py:367-403).  Bound in C (csrc/hb_pymarshal.c, dec_add): while
    # self._ch names an hb_dec and the column was received in place, the call is hb_dec_arrived1(self._ch, idx) and nothing else; a state
    # change is handed to _c_event, everything else to _add_slow.  Without the helper library the same dispatch in Python:
    def _add_py(self, idx, column=None):
        if column is None and self._ch is not None:
            st = self.ctx.lib.hb_dec_arrived1(self._cdec.h, idx)
            if st:
                self._c_event(st, idx)
            return
        return self._add_slow(idx, column)

    add = _marshal.as_method(_marshal.dec_add) if _marshal is not None and hasattr(_marshal, "dec_add") else _add_py

    def _fast_add(self):
        enough = len(self._available_points) >= self._min_points_required()
        if self._optimistic:
            if not enough or self._fast_optimistic():
                return
        if enough:
            self._fast_robust_update()
            if self._memo is not None:
                self._w_arm()

    def _catch_up_optimistic(self):
        """the plan-based optimistic path for a decoder that skipped it while the plan-free path looked available: guess from the
        first degree+1 arrivals, then every later arrival in order (reference :305-330)"""
        d = self.degree + 1
        later = self._z[d:]
        saved_z, saved_av = self._z, self._available_points
        self._z, self._available_points = saved_z[:d], set(saved_z[:d])
        self._decode_and_encode()
        for idx in later:
            self._z.append(idx)
            self._available_points.add(idx)
            if not self._optimistic_update(idx) or self.done():
                break
        self._z, self._available_points = saved_z, saved_av

    def pending(self):
        """defer_verdict: True while the quorum's launch is enqueued and nobody has waited for its verdict"""
        return self._pending

    def _settle(self):
        """wait for the pending verdict and act on it as the add() that completed the quorum would have; the senders announced meanwhile
        follow in order (an agreeing verdict ends the round: they are ignored, as the reference ignores arrivals after done)"""
        self._pending = False
        late, self._late = self._late, None
        self._c_event(self._cdec.settle(), None)
        for idx in late:
            if self._result is not None:
                break
            self.add(idx)

    def done(self):
        if self._pending:
            self._settle()
        return self._result is not None

    def get_results(self):
        if self._pending:
            self._settle()
        if self._result is not None:
            return self._result, self._confirmed_errors
        return None, None
