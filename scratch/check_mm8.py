"""Check + timing of the int8 matrix-core mat-vec (hb_mfma.hip) on its own, against Python integers."""
import ctypes
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def make(ctx, xs, d):
    lib = ctx.lib
    xh = ctx.host_elems(xs)
    h = ctypes.c_void_p()
    rc = lib.hb_debug_mm8_create(ctx.h, xh.ctypes.data, len(xs), d, ctypes.byref(h))
    assert rc == 0, rc
    return h


def apply(ctx, h, x_dev, d, n_out, chunks, out=None, layout="chunk", in_count=None):
    if out is None:
        out = ctx.empty(chunks * n_out)
    sc, sl = (n_out, 1) if layout == "chunk" else (1, chunks)
    rc = ctx.lib.hb_debug_mm8_apply(ctx.h, h, ctx.ptr(x_dev), d, 1, chunks * d if in_count is None else in_count, ctx.ptr(out), sc, sl,
                                    chunks * n_out, chunks, None, None)
    assert rc == 0, rc
    return out


def check(ctx, pts, d, chunks, seed, extreme=False, short=0):
    rng = random.Random(seed)
    n_out = len(pts)
    if extreme:
        pool = [0, 1, P - 1, (1 << 256) - 1, 1 << 255, int("80" * 32, 16), int("7f" * 32, 16)]
        xs = [rng.choice(pool) for _ in range(chunks * d)]
        arr = np.zeros((chunks * d, 4), dtype=np.uint64)
        for k, v in enumerate(xs):
            for j in range(4):
                arr[k, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        x_dev = ctx.to_device(arr)
    else:
        xs = [rng.randrange(P) for _ in range(chunks * d)]
        x_dev = ctx.upload_ints(xs)
    h = make(ctx, pts, d)
    in_count = chunks * d - short
    out = ctx.download_ints(apply(ctx, h, x_dev, d, n_out, chunks, in_count=in_count))
    bad = 0
    for c in range(chunks):
        for i in range(n_out):
            exp = sum(pow(pts[i], l, P) * (xs[c * d + l] if c * d + l < in_count else 0) for l in range(d)) % P
            if out[c * n_out + i] != exp:
                if bad < 5:
                    print("MISMATCH chunk", c, "row", i, hex(out[c * n_out + i]), hex(exp))
                bad += 1
    print(f"n_out={n_out} d={d} chunks={chunks} extreme={extreme} short={short}: {'OK' if bad == 0 else f'{bad} BAD'}")
    return bad == 0


def timing(ctx, n_out, d, reps=20, layout="col"):
    chunks = (1 << 20) // d + 1
    h = make(ctx, list(range(1, n_out + 1)), d)
    x_dev = ctx.empty(chunks * d)
    x_dev.random_(0, 1 << 62)
    out = ctx.empty(chunks * n_out)
    for _ in range(3):
        apply(ctx, h, x_dev, d, n_out, chunks, out, layout)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        apply(ctx, h, x_dev, d, n_out, chunks, out, layout)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"mm8 n={n_out} d={d} chunks={chunks} layout={layout}: {dt * 1e6:.1f} us  ({chunks * d / dt / 1e9:.2f} G shares/s)")


def main():
    ctx = Context.get(P)
    ok = True
    if "--time-only" not in sys.argv:
        ok &= check(ctx, list(range(1, 65)), 22, 37, 1)
        ok &= check(ctx, list(range(1, 65)), 22, 16, 2, extreme=True)
        ok &= check(ctx, list(range(1, 65)), 22, 19, 3, short=7)
        ok &= check(ctx, list(range(1, 17)), 6, 50, 6)
        ok &= check(ctx, list(range(3, 24)), 9, 19, 7)
        ok &= check(ctx, list(range(1, 65)), 22, 16 * 600 + 5, 8)
        ok &= check(ctx, list(range(1, 41)), 11, 16 * 700 + 3, 9)
    if not ok:
        sys.exit(1)
    timing(ctx, 64, 22, layout="chunk")
    timing(ctx, 64, 22, layout="col")


if __name__ == "__main__":
    main()
