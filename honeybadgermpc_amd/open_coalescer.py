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
k-th batch of one party meets the k-th batch of the others (tag ("coalesced", k)); see the class docstring for what
"same points" requires when several coroutines open concurrently (explicit flush() / max_pending_shares).  Concatenation needs no padding: a
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
    """awaitable result of one queued open; keeps its own value once delivered, so the coalescer can let go of the batch"""

    _UNSET = object()

    def __init__(self, owner, batch, index, length):
        self._owner, self._batch, self._index, self._length = owner, batch, index, length
        self._value = self._UNSET
        self._error = None

    def __await__(self):
        return self._get().__await__()

    async def _get(self):
        if self._error is not None:
            raise self._error
        if self._value is self._UNSET:
            try:
                self._value = await self._owner._result_of(self._batch, self._index)
            except HoneyBadgerMPCError as e:         # the batch itself failed: every await of this open reports it
                self._error = e
                raise
        return self._value

    def __del__(self):
        # an open nobody awaited must not pin its batch's tensors for the life of the program
        try:
            if self._value is self._UNSET and self._error is None:
                self._owner._delivered(self._batch, self._index)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class OpenCoalescer:
    """Where a batch is cut decides which opens travel together, and every party must cut at the same places:

      * `flush()` and `max_pending_shares` cut at points of the PROGRAM ORDER of open_share_array calls -- the order the
        reference already requires to be the same on every party (it numbers share ids by it, mpc.py:101-112).  These
        cuts are deterministic whatever the scheduler does; a program with several concurrent coroutines uses them
        (`cut_on_await=False`: awaiting an open whose batch was not cut raises instead of cutting).
      * `cut_on_await=True` (default, the convenient form of the docstring above): the first await of a pending open cuts
        its batch.  That point is part of the program order only if the opens of the batch were queued by the coroutine that
        awaits, with no other coroutine queueing in between; otherwise the composition of the batch would depend on message
        timing and differ between parties.  The coalescer checks this: a batch cut by an await must hold opens of ONE asyncio
        task, else it raises `RuntimeError` (loud and early instead of a hang in the exchange).

    Finished batches are dropped as soon as every open of theirs has been delivered (or abandoned): nothing accumulates
    over a long program."""

    def __init__(self, modulus, n, t, myid, get_send_recv, use_omega_powers=False, device=None, max_pending_shares=None, cut_on_await=True):
        self.ctx = Context.get(modulus, device)
        self.p, self.n, self.t, self.myid = modulus, n, t, myid
        self.get_send_recv = get_send_recv
        self.use_omega_powers = use_omega_powers
        self.max_pending_shares = max_pending_shares
        self.cut_on_await = bool(cut_on_await)
        self._queue = {}            # degree -> list of tensors pending in the current batch
        self._queued_by = {}        # degree -> set of asyncio tasks that queued into the current batch
        self._batch_id = {}         # degree -> id of the batch being filled
        self._tasks = {}            # (degree, batch id) -> task resolving to the list of per-open results (until all are delivered)
        self._undelivered = {}      # (degree, batch id) -> indices not yet delivered
        self._next_id = 0
        self._orphans = set()       # exchanges nobody waits for (strong references until they finish)
        self.opens, self.batches = 0, 0     # counters (diagnostics)

    @staticmethod
    def _current_task():
        try:
            return asyncio.current_task()
        except RuntimeError:
            return None

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
        self._queued_by.setdefault(degree, set()).add(self._current_task())
        batch = (degree, self._batch_id[degree])
        self._undelivered.setdefault(batch, set()).add(len(queue) - 1)
        pending = _PendingOpen(self, batch, len(queue) - 1, shares.shape[0])
        if self.max_pending_shares is not None and sum(s.shape[0] for s in queue) >= self.max_pending_shares:
            self._cut(degree)
        return pending

    def flush(self):
        """cut every batch that is being filled, at this point of the program order"""
        for degree in list(self._queue):
            self._cut(degree)

    def pending_batches(self):
        """batches still referenced (being filled, in flight, or with undelivered results): diagnostics / leak tests"""
        return len(self._tasks) + len(self._queue)

    # -- one batch = one batch_reconstruct_device -----------------------------------------------------------
    def _cut(self, degree):
        parts = self._queue.pop(degree, None)
        self._queued_by.pop(degree, None)
        if not parts:
            return
        bid = self._batch_id.pop(degree)
        self.batches += 1           # every party counts the same cuts, whoever is still interested in the results
        task = asyncio.ensure_future(self._run(parts, degree, bid))
        # a failed batch nobody awaits must not end as "Task exception was never retrieved"
        task.add_done_callback(lambda tk: tk.cancelled() or tk.exception())
        if self._undelivered.get((degree, bid)):
            self._tasks[(degree, bid)] = task
        else:
            # every open of this batch was dropped before the cut (garbage-collected, never awaited): the exchange still runs -- the other
            # parties count on this party's messages -- but nothing keeps its results
            self._undelivered.pop((degree, bid), None)
            self._orphans.add(task)
            task.add_done_callback(self._orphans.discard)

    async def _run(self, parts, degree, bid):
        torch = self.ctx.torch
        lengths = [int(s.shape[0]) for s in parts]
        merged = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        del parts
        send, recv = self.get_send_recv(("coalesced", degree, bid))
        opened = await batch_reconstruct_device(merged, self.p, self.t, self.n, self.myid, send, recv,
                                                use_omega_powers=self.use_omega_powers, degree=degree, device=self.ctx.device)
        if opened is None:
            raise HoneyBadgerMPCError("Batch reconstruction failed!")
        # clones: a delivered slice must not keep the whole batch's buffer alive
        return [piece.clone() for piece in torch.split(opened, lengths, dim=0)]

    async def _result_of(self, batch, index):
        degree, bid = batch
        if batch not in self._tasks and self._batch_id.get(degree) == bid:
            if not self.cut_on_await:
                raise RuntimeError("OpenCoalescer(cut_on_await=False): flush() (or max_pending_shares) must cut a batch before its opens are awaited")
            queued_by = self._queued_by.get(degree, set())
            me = self._current_task()
            if len(queued_by) > 1 or (queued_by and me not in queued_by):
                raise RuntimeError(
                    "OpenCoalescer: a batch cut by an await holds opens queued by more than one coroutine; where it is cut would "
                    "depend on scheduling and differ between parties.  Cut such batches with flush() or max_pending_shares "
                    "(cut_on_await=False makes this the rule).")
            self._cut(degree)                      # first await on this batch: everything queued so far goes out together
        task = self._tasks.get(batch)
        if task is None:
            raise RuntimeError("OpenCoalescer: this open's batch has been released (every open of it was delivered or abandoned)")
        try:
            # shield: a caller that gives up (asyncio.wait_for) cancels its own wait, not the exchange the other opens of the batch share
            results = await asyncio.shield(task)
        except asyncio.CancelledError:
            if task.cancelled():
                self._delivered(batch, index)      # the batch itself is gone for good
            raise                                  # only this wait was cancelled: the open stays undelivered and can be awaited again
        except BaseException:
            self._delivered(batch, index)          # failed for good: this open no longer pins the batch
            raise
        self._delivered(batch, index)              # handed over
        return results[index]

    def _delivered(self, batch, index):
        left = self._undelivered.get(batch)
        if left is None:
            return
        left.discard(index)
        if not left and batch in self._tasks:      # every open of a cut batch has been handed over: let go of its results
            del self._undelivered[batch]
            del self._tasks[batch]


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
