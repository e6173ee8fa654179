"""
Two-round batch opening of secret-shared values on the host API
(replaces honeybadgermpc/batch_reconstruction.py:25-227 -- same coroutine signature, same wire messages).

Wire protocol (what a peer running the reference sees):

    round "R1": party i sends ("R1", [P_c(x_j) for every chunk c]) to each party j      (reference :158-170)
    round "R2": party i sends ("R2", [P_c(0)  for every chunk c]) to every party        (reference :190-197)

where P_c is the polynomial whose degree+1 coefficients are chunk c of the caller's share vector.  Either round ends as soon
as an IncrementalDecoder is satisfied with the columns that have arrived (reference :43-61).  The codec objects are this
package's, so every encode, decode and robust decode of a round runs on the MI355X.

Structure (this file's own): `_Inbox` owns every background task of one open -- the tag router, one pump and n pending
receives per round -- and `_Open` runs the two rounds through one pair of helpers (`_scatter`, `_gather`).
"""
import asyncio
import logging
import random
import time

from .field import GF
from .polynomial import EvalPoint
from .reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory
from .utils.misc import chunk_data, flatten_lists, subscribe_recv, transpose_lists

ROUND_TAGS = ("R1", "R2")


async def fetch_one(awaitables):
    """Completion-order iterator over tasks: yields (position in `awaitables`, result).

    Reference :25-40.  Tasks that complete in the same loop iteration come out by position (the reference walks a set, whose
    order is the tasks' addresses)."""
    position = {}
    for i, task in enumerate(awaitables):
        position[task] = i
    waiting = set(position)
    while waiting:
        finished, waiting = await asyncio.wait(waiting, return_when=asyncio.FIRST_COMPLETED)
        for task in sorted(finished, key=position.__getitem__):
            yield position[task], task.result()


async def incremental_decode(receivers, encoder, decoder, robust_decoder, batch_size, t, degree, n):
    """One round's decode: columns go to an IncrementalDecoder in arrival order; its rows, or None when every column has
    arrived and it is still undecided (reference :43-61)."""
    state = IncrementalDecoder(encoder, decoder, robust_decoder, degree=degree, batch_size=batch_size, max_errors=t)
    async for sender, column in fetch_one(receivers):
        state.add(sender, column)
        if state.done():
            return state.get_results()[0]
    return None


def recv_each_party(recv, n):
    """Split a stream of (sender, payload) events by sender: (pump task, n `get` callables) (reference :64-85)."""
    per_party = tuple(asyncio.Queue() for _ in range(n))

    async def pump():
        while True:
            sender, payload = await recv()
            per_party[sender].put_nowait(payload)

    return asyncio.create_task(pump()), [box.get for box in per_party]


class _Inbox:
    """Every background task of one open.  Created in the order the reference creates them (router, then per round the pump
    and its n receives: reference :134-142), so the event loop schedules both implementations alike."""

    def __init__(self, recv, n):
        self.router, subscribe = subscribe_recv(recv)
        self.pumps, self.columns = [], {}
        for tag in ROUND_TAGS:
            pump, getters = recv_each_party(subscribe(tag), n)
            self.pumps.append(pump)
            self.columns[tag] = [asyncio.create_task(get()) for get in getters]

    def close(self):
        for task in (*self.pumps, self.router, *(c for tag in ROUND_TAGS for c in self.columns[tag])):
            task.cancel()


class _Open:
    """The two rounds of one party's open."""

    def __init__(self, field, t, n, degree, myid, send, inbox, use_omega_powers, robust_algorithm, num_chunks):
        self.t, self.n, self.degree, self.num_chunks = t, n, degree, num_chunks
        self.send, self.inbox = send, inbox
        self.timing = logging.LoggerAdapter(logging.getLogger("benchmark_logger"), {"node_id": myid})
        point = EvalPoint(field, n, use_omega_powers=use_omega_powers)
        family = Algorithm.FFT if use_omega_powers else Algorithm.VANDERMONDE
        self.encoder = EncoderFactory.get(point, family)
        self.decoder = DecoderFactory.get(point, family)
        self.robust = RobustDecoderFactory.get(t, point, algorithm=robust_algorithm)

    def _scatter(self, tag, messages):
        """messages[j] goes to party j (an empty batch has no columns and sends nothing in R1, as in the reference)."""
        began = time.time()
        for dest, message in enumerate(messages):
            self.send(dest, (tag, message))
        self.timing.info(f"[batch open] {tag} sent in {time.time() - began} s")

    async def _gather(self, tag):
        began = time.time()
        try:
            rows = await incremental_decode(self.inbox.columns[tag], self.encoder, self.decoder, self.robust,
                                            self.num_chunks, self.t, self.degree, self.n)
        except asyncio.CancelledError:
            # deliberate divergence: the reference swallows the cancellation and then reads a name it never bound
            # (reference :176-183); here the open's tasks are cancelled and the cancellation propagates
            self.inbox.close()
            raise
        if rows is None:
            logging.error(f"[batch open] {tag}: all columns in and no decision")
        else:
            self.timing.info(f"[batch open] {tag} decoded in {time.time() - began} s")
        return rows

    async def run(self, chunks):
        columns = transpose_lists(self.encoder.encode(chunks))          # column j is what party j must interpolate
        self._scatter("R1", columns)
        mine = await self._gather("R1")                                 # the polynomials this party is responsible for
        if mine is None:
            return None
        constants = [row[0] for row in mine]
        self._scatter("R2", [constants] * self.n)
        return await self._gather("R2")


async def batch_reconstruct(secret_shares, p, t, n, myid, send, recv, config=None,
                            use_omega_powers=False, debug=False, degree=None):
    """Open B shared secrets: `secret_shares` is party `myid`'s list of B GFElements; returns the B opened values as
    GFElements, or None when a round cannot decide (reference :88-227).  Shares travel in chunks of degree+1 (default t+1)."""
    degree = t if degree is None else degree
    field = GF(p)
    values = [share.value for share in secret_shares]
    if config is not None and config.induce_faults:
        logging.debug("[FAULT][batch open] this party sends random shares")
        values = [random.randint(0, p - 1) for _ in values]
    robust_algorithm = Algorithm.GAO if config is None else config.decoding_algorithm

    inbox = _Inbox(recv, n)
    chunks = chunk_data(values, degree + 1)
    opened = await _Open(field, t, n, degree, myid, send, inbox, use_omega_powers, robust_algorithm, len(chunks)).run(chunks)
    if opened is None:
        return None                                                      # as the reference: its tasks stay as they are
    inbox.close()
    flat = flatten_lists(opened)
    assert len(flat) >= len(values)
    return [field(v) for v in flat[: len(values)]]
