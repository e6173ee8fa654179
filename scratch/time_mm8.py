"""Timing-only driver for the int8-MFMA mat-vec at the config-3 encode shape (for rocprofv3)."""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scratch")
import test_mm8 as T  # noqa: E402

ctx = T.Context.get(T.P)
n_out, d = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 22
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
V = [[(i + 1) ** l for l in range(d)] for i in range(n_out)]
chunks = (1 << 20) // d + 1
h = T.make(ctx, V)
x = ctx.empty(chunks * d)
x.random_(0, 1 << 62)
out = ctx.empty(chunks * n_out)
for _ in range(3):
    T.apply(ctx, h, x, d, n_out, chunks, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    T.apply(ctx, h, x, d, n_out, chunks, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"mm8 n_out={n_out} d={d} chunks={chunks}: {dt * 1e6:.1f} us")
