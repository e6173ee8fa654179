"""
Two-round batch reconstruction of secret-shared values
(reference: honeybadgermpc/batch_reconstruction.py:25-227).

Same coroutine signature and wire messages as the reference:
("R1", column) to each party, ("R2", constant terms) to everyone.  The codec objects
(encoder / decoder / robust decoder / IncrementalDecoder) are this package's, so every
encode, decode and robust decode runs on the MI355X.
"""
import asyncio
import logging
import random
import time
from asyncio import Queue

from .field import GF
from .polynomial import EvalPoint
from .reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory
from .utils.misc import chunk_data, flatten_lists, subscribe_recv, transpose_lists


async def fetch_one(awaitables):
    """Yield (index, result) pairs in completion order (reference :25-40)."""
    index_of = {a: i for i, a in enumerate(awaitables)}
    pending = set(awaitables)
    while pending:
        done, pending = await asyncio.wait(pending, return_when=asyncio.FIRST_COMPLETED)
        for fut in done:
            yield (index_of[fut], await fut)


async def incremental_decode(receivers, encoder, decoder, robust_decoder, batch_size, t, degree, n):
    """Feed columns to an IncrementalDecoder as they arrive (reference :43-61)."""
    inc = IncrementalDecoder(encoder, decoder, robust_decoder, degree=degree, batch_size=batch_size, max_errors=t)
    async for idx, column in fetch_one(receivers):
        inc.add(idx, column)
        if inc.done():
            result, _ = inc.get_results()
            return result
    return None


def recv_each_party(recv, n):
    """Fan a (sender, payload) stream out into one queue per party (reference :64-85)."""
    queues = [Queue() for _ in range(n)]

    async def _pump():
        while True:
            j, o = await recv()
            queues[j].put_nowait(o)

    return asyncio.create_task(_pump()), [q.get for q in queues]


async def batch_reconstruct(secret_shares, p, t, n, myid, send, recv, config=None,
                            use_omega_powers=False, debug=False, degree=None):
    """Open B shared secrets held as `secret_shares` (GFElement list) by party `myid`.

    Returns the B reconstructed values as GFElements, or None when reconstruction fails.
    Reference :88-227; reconstruction proceeds in chunks of degree+1 values.
    """
    bench_logger = logging.LoggerAdapter(logging.getLogger("benchmark_logger"), {"node_id": myid})
    if degree is None:
        degree = t

    secret_shares = [v.value for v in secret_shares]
    if config is not None and config.induce_faults:
        logging.debug("[FAULT][BatchReconstruction] Sending random shares.")
        secret_shares = [random.randint(0, p - 1) for _ in range(len(secret_shares))]

    subscribe_task, subscribe = subscribe_recv(recv)
    del recv
    task_r1, recvs_r1 = recv_each_party(subscribe("R1"), n)
    data_r1 = [asyncio.create_task(r()) for r in recvs_r1]
    task_r2, recvs_r2 = recv_each_party(subscribe("R2"), n)
    data_r2 = [asyncio.create_task(r()) for r in recvs_r2]
    del subscribe
    background = [task_r1, task_r2, subscribe_task, *data_r1, *data_r2]

    def cancel_all():
        for task in background:
            task.cancel()

    fp = GF(p)
    decoding_algorithm = Algorithm.GAO if config is None else config.decoding_algorithm
    point = EvalPoint(fp, n, use_omega_powers=use_omega_powers)
    codec = Algorithm.FFT if use_omega_powers else Algorithm.VANDERMONDE
    enc = EncoderFactory.get(point, codec)
    dec = DecoderFactory.get(point, codec)
    robust_dec = RobustDecoderFactory.get(t, point, algorithm=decoding_algorithm)

    round1_chunks = chunk_data(secret_shares, degree + 1)
    num_chunks = len(round1_chunks)

    # R1: encode every chunk, send column j to party j
    start = time.time()
    encoded = enc.encode(round1_chunks)
    for dest, message in enumerate(transpose_lists(encoded)):
        send(dest, ("R1", message))
    bench_logger.info(f"[BatchReconstruct] P1 Send: {time.time() - start}")

    start = time.time()
    recons_r2 = None
    try:
        recons_r2 = await incremental_decode(data_r1, enc, dec, robust_dec, num_chunks, t, degree, n)
    except asyncio.CancelledError:
        # deliberate divergence: the reference swallows the cancellation and falls through with recons_r2 unbound
        # (batch_reconstruction.py:178-183); here the background tasks are cancelled and the cancellation propagates
        cancel_all()
        raise
    if recons_r2 is None:
        logging.error("[BatchReconstruct] P1 reconstruction failed!")
        return None
    bench_logger.info(f"[BatchReconstruct] P1 Reconstruct: {time.time() - start}")

    # R2: broadcast the constant terms
    start = time.time()
    message = [chunk[0] for chunk in recons_r2]
    for dest in range(n):
        send(dest, ("R2", message))
    bench_logger.info(f"[BatchReconstruct] P2 Send: {time.time() - start}")

    start = time.time()
    recons_p = None
    try:
        recons_p = await incremental_decode(data_r2, enc, dec, robust_dec, num_chunks, t, degree, n)
    except asyncio.CancelledError:
        cancel_all()
        raise
    if recons_p is None:
        logging.error("[BatchReconstruct] P2 reconstruction failed!")
        return None
    bench_logger.info(f"[BatchReconstruct] P2 Reconstruct: {time.time() - start}")

    cancel_all()
    result = flatten_lists(recons_p)
    assert len(result) >= len(secret_shares)
    return list(map(fp, result[: len(secret_shares)]))
