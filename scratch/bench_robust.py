"""Config 4: Welch-Berlekamp / Gao decode with t injected errors, n=100 t=33.
Generates codewords on the GPU path, corrupts exactly t random positions per codeword on the host
(bounded batch), decodes, verifies against the generating polynomials, reports codewords/s."""
import ctypes, random, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n, t = 100, 33
k = t + 1
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = Context.get(P); lib = ctx.lib
x = list(range(1, n + 1)); xh = ctx.host_elems(x)
gen = torch.Generator(device='cuda'); gen.manual_seed(4)
def rand(count):
    v = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device='cuda', generator=gen); v[:, 3] &= (1 << 61) - 1; return v
msg = rand(C * k)
code = ctx.empty(C * n)
ctx.check(lib.hb_vandermonde_batch_evaluate(ctx.h, np_ptr(xh), n, ctx.ptr(msg), C, k, ctx.ptr(code), ctx.stream()), "enc")
# corrupt exactly t positions per codeword with fresh random field elements
rng = np.random.default_rng(4)
pos = np.argsort(rng.random((C, n)), axis=1)[:, :t]
idx = torch.from_numpy((np.arange(C)[:, None] * n + pos).reshape(-1)).cuda()
bad = code.clone(); bad[idx] = rand(C * t)
present = torch.ones(C * n, dtype=torch.uint8, device='cuda')
for name in ("gao", "wb"):
    out = ctx.empty(C * k)
    t0 = None
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if name == "gao":
            err = ctx.empty(C * (n + 1)); elen = torch.zeros(C, dtype=torch.int32, device='cuda'); ok = torch.zeros(C, dtype=torch.uint8, device='cuda')
            ctx.check(lib.hb_gao_decode(ctx.h, np_ptr(xh), n, k, ctx.ptr(bad), C, ctx.ptr(out), ctx.ptr(err), ctx.ptr(elen), ctx.ptr(ok), ctx.stream()), "gao")
            good = bool(ok.all().item()) and bool((elen == t + 1).all().item())
        else:
            olen = torch.zeros(C, dtype=torch.int32, device='cuda'); st = torch.zeros(C, dtype=torch.int32, device='cuda')
            ctx.check(lib.hb_wb_decode(ctx.h, np_ptr(xh), n, k, ctx.ptr(bad), ctx.ptr(present), C, ctx.ptr(out), ctx.ptr(olen), ctx.ptr(st), ctx.stream()), "wb")
            good = bool((st == 0).all().item())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: C={C} n={n} t={t}: {dt*1e3:.1f} ms  {C/dt:.0f} codewords/s  all decoded={good}  coefficients == generating polynomials: {torch.equal(out, msg)}", flush=True)
