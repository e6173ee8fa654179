"""
Host-side reshapes and message plumbing around the kernels
(reference: honeybadgermpc/utils/misc.py:21-106; the TypeCheck decorator the reference
wraps these in is pure overhead and is not reproduced).
"""
import asyncio
from collections import defaultdict


def wrap_send(tag, send):
    """send(dest, msg) -> send(dest, (tag, msg))   (reference misc.py:21-29)."""
    if not isinstance(tag, str) or not callable(send):
        raise TypeError("wrap_send(tag: str, send: callable)")

    def _send(dest, message):
        send(dest, (tag, message))

    return _send


def chunk_data(data, chunk_size, default=0):
    """Split `data` into rows of `chunk_size`, padding the last row with `default`.

    Matches reference misc.py:33-51 including its quirk: an EMPTY input returns a
    flat list of `chunk_size` defaults, not a list holding one chunk."""
    if not isinstance(data, list) or not isinstance(chunk_size, int):
        raise TypeError("chunk_data(data: list, chunk_size: int)")
    if len(data) == 0:
        return [default] * chunk_size
    rows = [data[i : i + chunk_size] for i in range(0, len(data), chunk_size)]
    short = chunk_size - len(rows[-1])
    if short:
        rows[-1] = rows[-1] + [default] * short
    return rows


def flatten_lists(lists):
    """[[a, b], [c]] -> [a, b, c]   (reference misc.py:55-63)."""
    if not isinstance(lists, list):
        raise TypeError("flatten_lists(lists: list)")
    flat = []
    for inner in lists:
        flat += inner
    return flat


def transpose_lists(lists):
    """Row-major 2-D list transpose (reference misc.py:67-73)."""
    if not isinstance(lists, list):
        raise TypeError("transpose_lists(lists: list)")
    width = len(lists[0])  # like the reference, the first row fixes the width (IndexError on [])
    return [[row[i] for row in lists] for i in range(width)]


def subscribe_recv(recv):
    """Demultiplex `(sender, (tag, payload))` events into one queue per tag
    (reference misc.py:76-106).  Returns (background_task, subscribe)."""
    queues = defaultdict(asyncio.Queue)
    claimed = set()

    async def _pump():
        while True:
            sender, (tag, payload) = await recv()
            queues[tag].put_nowait((sender, payload))

    def subscribe(tag):
        assert tag not in claimed, f"tag {tag!r} subscribed twice"
        claimed.add(tag)
        return queues[tag].get

    return asyncio.create_task(_pump()), subscribe
