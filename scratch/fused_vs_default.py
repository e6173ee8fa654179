"""Small-entry plans: the reference-shaped validation on k_mm8 against the fused decode + validate on k_mm8w.
usage: python scratch/fused_vs_default.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import BatchOpen
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P)
g = torch.Generator(device="cuda"); g.manual_seed(5)
for n, t in [(8, 2), (8, 3), (16, 5), (24, 7), (32, 10), (48, 15), (64, 21), (96, 20), (100, 9)]:
    d = t + 1
    B = (1 << 20) // d * d
    C = B // d
    op = BatchOpen(P, n, t, z=list(range(d)), zc=list(range(d, min(n, d + t))), max_shares=B)
    if not op.uses_matrix_cores():
        print(n, t, "not on the matrix cores"); continue
    sh = torch.randint(0, (1 << 62), (B, 4), dtype=torch.int64, device="cuda", generator=g)
    cols = op.r1_encode(sh)
    res = {}
    for name, fused, arrived in (("full re-encode", False, False), ("compared rows only (k_mm8 family)", False, True), ("fused (k_mm8w)", True, False)):
        op.set_fused_validate(fused)
        op.set_validate_arrived_only(arrived)
        if fused and not op.uses_fused_validate():
            res[name] = None; continue
        for _ in range(3):
            op.r1_decode(cols, B); out = op.r2_decode(cols, B)
        assert op.ok() and torch.equal(out, sh)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            op.r1_decode(cols, B); op.r2_decode(cols, B)
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"n={n} t={t}: two decodes + validation, ms: " + ", ".join(f"{k} {v if v is None else round(v, 3)}" for k, v in res.items()))
