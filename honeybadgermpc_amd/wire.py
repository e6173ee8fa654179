"""
Packed wire / disk format for field elements.

The reference ships R1/R2 messages as pickled lists of Python ints (ipc.py:111, mpc.py:196)
and stores shares as decimal text (preprocessing.py:106-169); both must be re-marshalled
element by element on every hop.  The kernels' native layout -- 32 bytes per element,
little-endian, canonical -- is already a wire format: a message is the raw bytes of a
device buffer, prefixed by a 16-byte header.

    magic  b"HBFE"   4 bytes
    limbs  u32 LE    4 bytes   (4 for p < 2^256, 1 for p < 2^64)
    count  u64 LE    8 bytes
    data   count * limbs * 8 bytes
"""
import struct

import numpy as np

MAGIC = b"HBFE"
HEADER = struct.Struct("<4sIQ")


def pack_limbs(limbs):
    """(count, n_limbs) uint64 ndarray -> bytes"""
    a = np.ascontiguousarray(limbs, dtype=np.uint64)
    if a.ndim != 2:
        raise ValueError("expected a (count, limbs) array")
    return HEADER.pack(MAGIC, a.shape[1], a.shape[0]) + a.tobytes()


def unpack_limbs(blob):
    """bytes -> (count, n_limbs) uint64 ndarray (a view on the message body)"""
    if len(blob) < HEADER.size:
        raise ValueError("truncated message")
    magic, limbs, count = HEADER.unpack_from(blob, 0)
    if magic != MAGIC or limbs not in (1, 4):
        raise ValueError("not a packed field-element message")
    need = HEADER.size + count * limbs * 8
    if len(blob) != need:
        raise ValueError(f"message length {len(blob)} != {need}")
    return np.frombuffer(blob, dtype=np.uint64, offset=HEADER.size).reshape(count, limbs)


def pack_ints(values, modulus):
    """list[int] -> bytes (values reduced mod p)"""
    from ._capi import ints_to_limbs

    return pack_limbs(ints_to_limbs(list(values), modulus, 32))


def unpack_ints(blob):
    from ._capi import limbs_to_ints

    a = unpack_limbs(blob)
    return limbs_to_ints(a, a.shape[1] * 8)


def tensor_to_wire(t):
    """device / host int64 tensor (count, limbs) -> bytes (one D2H copy, no per-element work)"""
    return pack_limbs(t.detach().cpu().numpy().view(np.uint64))


def wire_to_tensor(blob, device=None):
    import torch

    a = unpack_limbs(blob)
    t = torch.from_numpy(a.view(np.int64).copy())
    return t.to(device) if device is not None else t
