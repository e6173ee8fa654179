"""fft of one big polynomial (the reference's benchmark sizes, benchmark/test_benchmark_polynomial.py:22-48): the four-step
transform over the LDS kernel against the stage-by-stage loop (HB_NTT_STAGE_LOOP=1), packed device tensors in and out."""
import os, sys, time
import torch
sys.path.insert(0, '.')
from honeybadgermpc_amd import ntl
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.field import GF
from honeybadgermpc_amd.polynomial import get_omega
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P)
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
for logn in (12, 14, 16, 18, 20):
    n = 1 << logn
    omega = get_omega(GF(P), n, seed=0).value
    co = torch.randint(-(1 << 63), (1 << 63) - 1, (1, n, 4), dtype=torch.int64, device='cuda', generator=gen); co[:, :, 3] &= (1 << 61) - 1
    res = {}
    for mode in ("four-step", "stage loop"):
        if mode == "stage loop": os.environ["HB_NTT_STAGE_LOOP"] = "1"
        else: os.environ.pop("HB_NTT_STAGE_LOOP", None)
        out = ntl.fft_batch_evaluate(co, omega, P, n, n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            out = ntl.fft_batch_evaluate(co, omega, P, n, n)
        torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 5, out)
    same = torch.equal(res["four-step"][1], res["stage loop"][1])
    print(f"n = 2^{logn}: four-step {res['four-step'][0]*1e3:.3f} ms, stage loop {res['stage loop'][0]*1e3:.3f} ms, identical {same}", flush=True)
