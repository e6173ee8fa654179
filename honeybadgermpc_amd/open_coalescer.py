"""
Device-side `Mpc.open_share_array` (reference honeybadgermpc/mpc.py:164-219).

The reference opens every ShareArray with its own batch_reconstruct (its own share id, its own R1 / R2 exchange):
an MPC program that opens 64 small arrays in one step (Beaver multiplications, progs/mixins/share_arithmetic.py:24-45)
pays 64 two-round exchanges and, on a GPU, 64 sets of launches over a few chunks each.  `OpenCoalescer` merges what is
pending into ONE open:

    co = OpenCoalescer(p, n, t, myid, get_send_recv)          # get_send_recv(tag) -> (send, recv), as the runtime hands out
    a = co.open_share_array(x_shares)                          # (len, limbs) tensors; returns an awaitable at once
    b = co.open_share_array(y_shares)
    ...
    xs = await a                                               # the first await cuts the batch: everything queued so far
    ys = await b                                               # travels as one concatenated share vector

Every party runs the same program, so every party cuts its batches at the same points of the program order and the
k-th batch of one party meets the k-th batch of the others (tag ("coalesced", k)).  Concatenation needs no padding: a
reconstruction is element-wise in the share vector, chunk boundaries carry no meaning (batch_reconstruction.py:158,223-227).
An empty array resolves to an empty tensor without joining a batch (mpc.py:175-177).

`robust_reconstruct_device` is the single-share robust open (robust_reconstruction.py:14-30) on the device decoder.
"""
import asyncio

from ._capi import Context
from .batch_reconstruction import fetch_one
from .device import DeviceIncrementalDecoder
from .device_reconstruction import batch_reconstruct_device


class HoneyBadgerMPCError(Exception):
    """as the reference raises when a batch reconstruction fails (mpc.py:183-186)"""


class _PendingOpen:
    def __init__(self, owner, batch, index, length):
        self._owner, self._batch, self._index, self._length = owner, batch, index, length

    def __await__(self):
        return self._owner._result_of(self._batch, self._index).__await__()


class OpenCoalescer:
    def __init__(self, modulus, n, t, myid, get_send_recv, use_omega_powers=False, device=None, max_pending_shares=None):
        self.ctx = Context.get(modulus, device)
        self.p, self.n, self.t, self.myid = modulus, n, t, myid
        self.get_send_recv = get_send_recv
        self.use_omega_powers = use_omega_powers
        self.max_pending_shares = max_pending_shares
        self._queue = {}            # degree -> list of tensors pending in the current batch
        self._batch_id = {}         # degree -> id of the batch being filled
        self._tasks = {}            # (degree, batch id) -> task resolving to the list of per-open results
        self._next_id = 0
        self.opens, self.batches = 0, 0     # counters (diagnostics)

    # -- queueing ----------------------------------------------------------------------------------------
    def open_share_array(self, shares, degree=None):
        """shares: (len, limbs) limb tensor on this context's device (or a list of ints).  Returns an awaitable that yields
        the (len, limbs) tensor of opened values; raises HoneyBadgerMPCError if the reconstruction fails."""
        if not hasattr(shares, "shape"):
            shares = self.ctx.upload_ints(list(shares))
        shares = self.ctx.elems(shares, what="shares")
        degree = self.t if degree is None else degree
        self.opens += 1
        if shares.shape[0] == 0:
            done = asyncio.get_event_loop().create_future()
            done.set_result(shares)
            return done
        queue = self._queue.setdefault(degree, [])
        if degree not in self._batch_id:
            self._batch_id[degree] = self._next_id
            self._next_id += 1
        queue.append(shares)
        pending = _PendingOpen(self, (degree, self._batch_id[degree]), len(queue) - 1, shares.shape[0])
        if self.max_pending_shares is not None and sum(s.shape[0] for s in queue) >= self.max_pending_shares:
            self._cut(degree)
        return pending

    def flush(self):
        """cut every batch that is being filled (the first await of one of its opens does the same)"""
        for degree in list(self._queue):
            self._cut(degree)

    # -- one batch = one batch_reconstruct_device -----------------------------------------------------------
    def _cut(self, degree):
        parts = self._queue.pop(degree, None)
        if not parts:
            return
        bid = self._batch_id.pop(degree)
        self.batches += 1
        self._tasks[(degree, bid)] = asyncio.ensure_future(self._run(parts, degree, bid))

    async def _run(self, parts, degree, bid):
        torch = self.ctx.torch
        lengths = [int(s.shape[0]) for s in parts]
        merged = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        send, recv = self.get_send_recv(("coalesced", degree, bid))
        opened = await batch_reconstruct_device(merged, self.p, self.t, self.n, self.myid, send, recv,
                                                use_omega_powers=self.use_omega_powers, degree=degree, device=self.ctx.device)
        if opened is None:
            raise HoneyBadgerMPCError("Batch reconstruction failed!")
        return list(torch.split(opened, lengths, dim=0))

    async def _result_of(self, batch, index):
        degree, bid = batch
        if batch not in self._tasks and self._batch_id.get(degree) == bid:
            self._cut(degree)                      # first await on this batch: everything queued so far goes out together
        results = await self._tasks[batch]
        return results[index]


async def robust_reconstruct_device(column_futures, modulus, n, t, degree=None, use_omega_powers=False, device=None):
    """Single-share robust open on the device decoder (reference robust_reconstruction.py:14-30, batch size 1).
    column_futures[i] resolves to party i's share: an int, a GFElement, or a (1, limbs) tensor.
    -> ((degree+1, limbs) coefficient tensor, set of erroneous senders) or (None, None)."""
    inc = DeviceIncrementalDecoder(modulus, n, t, degree=degree, batch_size=1, use_omega_powers=use_omega_powers, device=device)
    async for idx, value in fetch_one(column_futures):
        if hasattr(value, "value"):
            value = value.value
        inc.add(idx, [value] if isinstance(value, int) else value)
        if inc.done():
            coeffs, errors = inc.get_results()
            return coeffs[0], errors
    return None, None
