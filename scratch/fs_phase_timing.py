"""Where the waves of config 3's three launches spend their cycles (s_memtime around the phases): needs the timing build,
scratch/build_variant2.sh timing "hb_mfma_fused hb_mfma" -DHB_FS_TIMING -DHB_MM8_TIMING;  HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_timing.so"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen  # noqa: E402

FS = ["prologue+unit 0 scaled", "barrier wait", "scale next unit", "MFMA half 0", "park half 0", "MFMA half 1", "words+reduce+store/compare", "lgkm wait before barrier"]
MM = ["prologue", "barrier wait", "DMA issue", "MFMA half 0", "park half 0", "MFMA half 1", "vmcnt wait", "words+reduce+store"]


def table(t, names, title, by_group=None):
    tot = t.sum(axis=1)
    print(f"{title}: per-wave total ticks mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
    for k, nm in enumerate(names):
        col = t[:, k]
        extra = ""
        if by_group is not None:
            extra = "   " + "  ".join(f"{g}: {col[idx].mean():8.0f}" for g, idx in by_group.items())
        print(f"  {nm:28s} {col.mean():9.0f} ticks/wave  {100 * col.mean() / tot.mean():5.1f} %  (min {col.min():.0f}, max {col.max():.0f}){extra}")


def main():
    n, t, B = 64, 21, 1 << 20
    d = t + 1
    C = (B + d - 1) // d
    ctx = Context.get(bench.BLS, 0)
    lib = ctx.lib
    shares0, r1_cols, r2_cols, secrets, x = bench.make_inputs(torch, ctx, n, t, B, False, seed=1000)
    order = np.random.Generator(np.random.PCG64(2024)).permutation(n).tolist()
    op = BatchOpen(bench.BLS, n, t, z=order[:d], zc=order[d:d + t], max_shares=B, device=0)
    if op.uses_fused_validate():
        op.set_fused_validate(True)
    r1_out, r2_msg, result = ctx.empty(n * C), ctx.empty(C), ctx.empty(B)
    for _ in range(200):
        op.r1_encode(shares0, out=r1_out); op.r1_decode(r1_cols, B, out=r2_msg); op.r2_decode(r2_cols, B, out=result)
    torch.cuda.synchronize()
    fs = lib.hb_debug_fs_timing
    fs.argtypes = [ctypes.c_void_p, ctypes.c_int]
    mm = lib.hb_debug_mm8_timing
    mm.argtypes = [ctypes.c_void_p, ctypes.c_int]
    waves = np.arange(256 * 8) % 8
    groups = {"waves 0-3": waves < 4, "waves 4-7": waves >= 4}
    for name, leg in (("R1 decode+validate (k_mm8f)", lambda: op.r1_decode(r1_cols, B, out=r2_msg)), ("R2 decode+validate (k_mm8f)", lambda: op.r2_decode(r2_cols, B, out=result))):
        for _ in range(3):
            op.r1_encode(shares0, out=r1_out); leg()
        torch.cuda.synchronize()
        buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
        assert fs(buf.ctypes.data, buf.size) == 0
        table(buf.reshape(256 * 8, 8).astype(np.float64), FS, name, groups)
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    assert mm(buf.ctypes.data, buf.size) == 0
    table(buf.reshape(2048, 8).astype(np.float64), MM, "R1 encode (k_mm8)")


main()
