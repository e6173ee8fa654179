"""The same per-party open at config 3 measured at the two outer boundaries (neither is bench.py's `value`):
  (a) host buffers: inputs start in pinned host memory, results end there (PCIe both ways, BatchOpen in between);
  (b) the reference's own boundary: lists of Python ints through the ntl drop-in (3 evaluate + 2 interpolate calls)."""
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from honeybadgermpc_amd import ntl  # noqa: E402
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    n, t, b = 64, 21, 1 << 20
    d = t + 1
    c = (b + d - 1) // d
    ctx = Context.get(P)
    rnd = random.Random(1)
    z = list(range(d))
    op = BatchOpen(P, n, t, z=z, zc=list(range(d, d + t)), max_shares=b)
    shares = ctx.empty(b)
    shares.random_(0, 1 << 62)
    r1 = op.r1_encode(shares)                                  # consistent columns: everybody holds the same vector here
    # (a) pinned host <-> device
    h_sh = shares.cpu().pin_memory()
    h_cols = r1.cpu().pin_memory()
    h_r1 = torch.empty((n * c, 4), dtype=torch.int64).pin_memory()
    h_msg = torch.empty((c, 4), dtype=torch.int64).pin_memory()
    h_res = torch.empty((b, 4), dtype=torch.int64).pin_memory()
    d_cols = ctx.empty(n * c)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dsh = h_sh.to(ctx.tdev, non_blocking=True)
        h_r1.copy_(op.r1_encode(dsh), non_blocking=True)
        d_cols.copy_(h_cols, non_blocking=True)
        h_msg.copy_(op.r1_decode(d_cols, b), non_blocking=True)
        d_cols.copy_(h_cols, non_blocking=True)
        h_res.copy_(op.r2_decode(d_cols, b), non_blocking=True)
        assert op.ok()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    moved = (b + n * c + n * c + c + n * c + b) * 32
    print(f"(a) pinned host buffers in and out: {dt * 1e3:.2f} ms per open = {b / dt / 1e6:.0f} M shares/s ({moved / 1e6:.0f} MB over PCIe, {moved / dt / 1e9:.1f} GB/s)")
    # (b) lists of Python ints
    xs = list(range(1, n + 1))
    polys = [[rnd.randrange(P) for _ in range(d)] for _ in range(c)]
    t0 = time.perf_counter()
    enc = ntl.vandermonde_batch_evaluate(xs, polys, P)
    t1 = time.perf_counter()
    rows = [row[:d] for row in enc]
    t2 = time.perf_counter()
    dec = ntl.vandermonde_batch_interpolate(xs[:d], rows, P)
    t3 = time.perf_counter()
    assert dec == polys
    ev, it = t1 - t0, t3 - t2
    total = 3 * ev + 2 * it
    print(f"(b) list-of-int boundary: evaluate {ev * 1e3:.0f} ms, interpolate {it * 1e3:.0f} ms per call -> 3 + 2 calls = {total * 1e3:.0f} ms per open = {b / total / 1e6:.2f} M shares/s")
    # (c) the same drop-in functions with packed batches: numpy uint64 (rows, width, 4) in and out (host memory, pageable),
    #     and torch device tensors in and out
    import oracle  # noqa: F401  (not used: only here to make clear nothing below touches it)
    pn = np.frombuffer(b"".join(v.to_bytes(32, "little") for row in polys for v in row), dtype=np.uint64).reshape(c, d, 4).copy()
    for kind, give in (("numpy (pageable host)", lambda a: a), ("torch device tensor", lambda a: torch.from_numpy(a.view(np.int64)).cuda())):
        arr = give(pn)
        ev = it = None
        for _ in range(4):                               # the first calls build the tables; steady state = the fastest
            torch.cuda.synchronize(); t0 = time.perf_counter()
            enc_p = ntl.vandermonde_batch_evaluate(xs, arr, P)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            rows_p = enc_p[:, :d].copy() if isinstance(enc_p, np.ndarray) else enc_p[:, :d].contiguous()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            dec_p = ntl.vandermonde_batch_interpolate(xs[:d], rows_p, P)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            ev = t1 - t0 if ev is None else min(ev, t1 - t0)
            it = t3 - t2 if it is None else min(it, t3 - t2)
        same = np.array_equal(dec_p if isinstance(dec_p, np.ndarray) else dec_p.cpu().numpy().view(np.uint64), pn)
        total = 3 * ev + 2 * it
        print(f"(c) packed batches, {kind}: evaluate {ev * 1e3:.2f} ms, interpolate {it * 1e3:.2f} ms per call -> 3 + 2 calls = {total * 1e3:.2f} ms per open = {b / total / 1e6:.0f} M shares/s, round trip exact: {same}")


main()
