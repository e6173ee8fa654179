"""Prototype check + timing of the int8-MFMA mat-vec (hb_mm8_*) against Python integers."""
import ctypes
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context, ints_to_limbs  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def balanced_limbs(v, n=16):
    out = []
    for _ in range(n):
        dgt = ((v + 128) % 256) - 128
        out.append(dgt)
        v = (v - dgt) // 256
    assert v == 0, "entry too large"
    return out


def words(v, n):
    return [(v >> (32 * j)) & 0xFFFFFFFF for j in range(n)]


def make(ctx, M):
    n_out, d = len(M), len(M[0])
    limbs = np.zeros((n_out, d, 16), dtype=np.int8)
    for i in range(n_out):
        for l in range(d):
            limbs[i, l, :] = balanced_limbs(M[i][l])
    bias = sum(128 << (8 * a) for a in range(32))
    mu = (1 << 406) // P
    mu6 = np.array([(mu >> (29 * k)) & ((1 << 29) - 1) for k in range(6)], dtype=np.uint32)
    assert mu >> (29 * 6) == 0
    # second-cut kernel: digit-form row constant with the accumulator bias folded in
    BIAS = 5800000
    btot = sum(BIAS << (8 * c) for c in range(47))
    crowd = np.zeros((n_out, 16), dtype=np.uint32)
    for i in range(n_out):
        c = (bias * sum(M[i])) % P + (P << 137) - btot
        assert 0 <= c < 1 << 393
        crowd[i, :14] = [(c >> (29 * k)) & ((1 << 29) - 1) for k in range(14)]
    h = ctypes.c_void_p()
    lib = ctx.lib
    lib.hb_mm8_create.restype = ctypes.c_int
    lib.hb_mm8_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.hb_mm8_create(ctx.h, n_out, d, limbs.ctypes.data, crowd.ctypes.data, mu6.ctypes.data, ctypes.byref(h))
    assert rc == 0, rc
    lib.hb_mm8_apply.restype = ctypes.c_int
    lib.hb_mm8_apply.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    return h


def apply(ctx, h, x_dev, d, n_out, chunks, out=None):
    if out is None:
        out = ctx.empty(chunks * n_out)
    rc = ctx.lib.hb_mm8_apply(ctx.h, h, ctx.ptr(x_dev), d, 1, chunks * d, ctx.ptr(out), n_out, 1, chunks * n_out, chunks)
    assert rc == 0, rc
    return out


def check(ctx, M, chunks, seed, extreme=False):
    rng = random.Random(seed)
    n_out, d = len(M), len(M[0])
    if extreme:
        pool = [0, 1, P - 1, (1 << 256) - 1, 1 << 255, 0x8080808080808080808080808080808080808080808080808080808080808080, 0x7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F]
        xs = [rng.choice(pool) for _ in range(chunks * d)]
        arr = np.zeros((chunks * d, 4), dtype=np.uint64)
        for k, v in enumerate(xs):
            for j in range(4):
                arr[k, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        x_dev = ctx.to_device(arr)
    else:
        xs = [rng.randrange(P) for _ in range(chunks * d)]
        x_dev = ctx.upload_ints(xs)
    h = make(ctx, M)
    out = ctx.download_ints(apply(ctx, h, x_dev, d, n_out, chunks))
    bad = 0
    for c in range(chunks):
        for i in range(n_out):
            exp = sum(M[i][l] * xs[c * d + l] for l in range(d)) % P
            if out[c * n_out + i] != exp:
                if bad < 5:
                    print("MISMATCH chunk", c, "row", i, hex(out[c * n_out + i]), hex(exp))
                bad += 1
    print(f"n_out={n_out} d={d} chunks={chunks} extreme={extreme}: {'OK' if bad == 0 else f'{bad} BAD'}")
    return bad == 0


def main():
    ctx = Context.get(P)
    ok = True
    V = [[(i + 1) ** l for l in range(22)] for i in range(64)]
    ok &= check(ctx, V, 37, 1)
    ok &= check(ctx, V, 16, 2, extreme=True)
    rng = random.Random(5)
    S = [[rng.randrange(-(1 << 125), 1 << 125) for l in range(22)] for i in range(22)]
    ok &= check(ctx, S, 33, 3)
    ok &= check(ctx, S, 16, 4, extreme=True)
    V2 = [[(i + 1) ** l for l in range(6)] for i in range(16)]
    ok &= check(ctx, V2, 50, 6)
    V3 = [[(i + 3) ** l for l in range(9)] for i in range(21)]
    ok &= check(ctx, V3, 19, 7)
    ok &= check(ctx, V, 16 * 600 + 5, 8)
    V4 = [[(i + 1) ** l for l in range(11)] for i in range(40)]
    ok &= check(ctx, V4, 16 * 700 + 3, 9)
    ok &= check(ctx, S, 16 * 1100 + 1, 10)
    if not ok:
        sys.exit(1)
    # timing at the config-3 shape
    chunks = (1 << 20) // 22 + 1
    h = make(ctx, V)
    x_dev = ctx.empty(chunks * 22)
    x_dev.random_(0, 1 << 62)
    for _ in range(3):
        apply(ctx, h, x_dev, 22, 64, chunks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = ctx.empty(chunks * 64)
    reps = 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        apply(ctx, h, x_dev, 22, 64, chunks, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    macs = (chunks / 16) * 4 * 282 * 16 * 16 * 64  # passes (16 rows x 16 chunks) x 282 MFMAs x MACs each
    print(f"mm8 encode n=64 d=22 chunks={chunks}: {dt * 1e6:.1f} us  ({chunks * 22 / dt / 1e9:.2f} G shares/s, {2 * macs / dt / 1e12:.0f} TOPS int8)")


if __name__ == "__main__":
    main()
