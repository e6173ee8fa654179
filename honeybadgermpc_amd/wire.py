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
    """bytes -> (count, n_limbs) uint64 ndarray (a view on the message body).
    Anything that is not a well-formed message -- including a payload that is not bytes-like at all, which an
    untrusted peer behind an unpickling transport can send -- raises ValueError."""
    if not isinstance(blob, (bytes, bytearray, memoryview)):
        raise ValueError(f"packed message must be bytes-like, got {type(blob).__name__}")
    if isinstance(blob, memoryview):
        blob = blob.tobytes()
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


def wire_to_tensor(blob, device=None, out=None):
    """bytes -> (count, limbs) int64 tensor.  out: a tensor of that shape to receive into (a row of a decoder's party-major buffer:
    one H2D copy straight to where the kernels read it); a payload of another shape raises ValueError."""
    import torch

    a = unpack_limbs(blob)
    t = torch.from_numpy(a.view(np.int64).copy())
    if out is not None:
        if tuple(out.shape) != tuple(t.shape):
            raise ValueError("payload does not have the receiving buffer's shape")
        out.copy_(t, non_blocking=False)
        return out
    return t.to(device) if device is not None else t


# ---- share files (reference preprocessing.py:106-169) -----------------------------------------------
# The reference's `.share` file is decimal text, one integer per line: modulus, degree, party id, then the
# values.  read_share_file / write_share_file speak that format (interoperability with files the reference's
# dealer wrote); the *_packed variants keep the three metadata fields but store the values in the kernels' layout,
# so that a file is read straight into a device buffer.
SHARE_MAGIC = b"HBSH"
SHARE_HEADER = struct.Struct("<4sI32sqq")       # magic, limbs, modulus (32 bytes LE), degree, party id


def share_filename(prefix, n, t, context_id):
    """reference build_filename (preprocessing.py:150-169)"""
    return f"{prefix}_{n}_{t}-{context_id}.share"


def write_share_file(file_name, modulus, degree, context_id, values, append=False):
    """The reference's text format (preprocessing.py:125-148), including its append rule: appending to an existing file
    requires identical metadata; appending to a missing file creates it."""
    import os

    if not os.path.isfile(file_name):
        append = False
    if append:
        with open(file_name, "r") as f:
            meta = tuple(int(f.readline()) for _ in range(3))
        expected = (modulus, degree, context_id)
        assert meta == expected, f"File {file_name} expected to have metadata {expected}, but had {meta}"
    with open(file_name, "a" if append else "w") as f:
        if not append:
            print(modulus, degree, context_id, file=f, sep="\n")
        print(*values, file=f, sep="\n")


def read_share_file(file_name, modulus):
    """-> (degree, context_id, values) from the reference's text format (preprocessing.py:106-123)."""
    with open(file_name, "r") as f:
        values = list(map(int, f.read().splitlines()))
    assert len(values) >= 3
    assert values[0] == modulus, f"Expected file to have modulus {modulus}, but found {values[0]}"
    return values[1], values[2], values[3:]


def write_share_file_packed(file_name, modulus, degree, context_id, limbs, append=False):
    """limbs: (count, 4) uint64 ndarray or int64 tensor (any device).  Same metadata and append rule as the text format."""
    import os

    if hasattr(limbs, "detach"):
        limbs = limbs.detach().cpu().numpy().view(np.uint64)
    a = np.ascontiguousarray(limbs, dtype=np.uint64)
    if a.ndim != 2:
        raise ValueError("expected a (count, limbs) array")
    header = SHARE_HEADER.pack(SHARE_MAGIC, a.shape[1], int(modulus).to_bytes(32, "little"), degree, context_id)
    if not os.path.isfile(file_name):
        append = False
    if append:
        with open(file_name, "rb") as f:
            have = f.read(SHARE_HEADER.size)
        assert have == header, f"File {file_name} has different metadata"
    with open(file_name, "ab" if append else "wb") as f:
        if not append:
            f.write(header)
        f.write(a.tobytes())


def read_share_file_packed(file_name, modulus, device=None):
    """-> (degree, context_id, limbs); limbs is a (count, n_limbs) uint64 ndarray, or an int64 tensor on `device`."""
    with open(file_name, "rb") as f:
        blob = f.read()
    if len(blob) < SHARE_HEADER.size:
        raise ValueError("truncated share file")
    magic, limbs, mod, degree, context_id = SHARE_HEADER.unpack_from(blob, 0)
    if magic != SHARE_MAGIC or limbs not in (1, 4):
        raise ValueError("not a packed share file")
    assert int.from_bytes(mod, "little") == modulus, f"Expected file to have modulus {modulus}"
    body = len(blob) - SHARE_HEADER.size
    if body % (limbs * 8):
        raise ValueError("share file length is not a whole number of elements")
    a = np.frombuffer(blob, dtype=np.uint64, offset=SHARE_HEADER.size).reshape(-1, limbs)
    if device is None:
        return degree, context_id, a
    import torch

    return degree, context_id, torch.from_numpy(a.view(np.int64).copy()).to(device)
