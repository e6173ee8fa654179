import ctypes, sys, random
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.field import GF
from honeybadgermpc_amd.polynomial import EvalPoint
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P); lib = ctx.lib
n, t, c = 64, 21, 40; d = t + 1
pt = EvalPoint(GF(P), n, use_omega_powers=True)
x = [pt(i).value for i in range(n)]
rnd = random.Random(5)
polys = [[rnd.randrange(P) for _ in range(d)] for _ in range(c)]
enc = oracle.vandermonde_batch_evaluate(x, polys, P)
cols = ctx.upload_ints([enc[k][j] for j in range(n) for k in range(c)])
h = ctypes.c_void_p()
assert lib.hb_quick_dec_create(ctx.h, np_ptr(ctx.host_elems(x)), n, ctypes.byref(h), ctx.stream()) == 0
order = list(range(n)); rnd.shuffle(order)
z = np.array(order[:d], dtype=np.int32)
for nc in (1, 2, 5, 12, 21):
    zc = np.array(order[d:d + nc], dtype=np.int32)
    for n_coef in (d, 1):
        assert lib.hb_quick_dec_arrivals(h, np_ptr(z), d, nc, n_coef, ctx.stream()) == 0
        out = ctx.empty(c * n_coef); out.zero_()
        flag, first = ctypes.c_int32(-1), ctypes.c_int32(-1)
        rc = lib.hb_quick_dec_decide(h, np_ptr(zc), nc, ctx.ptr(cols), c, 0, c, ctx.ptr(out), ctypes.byref(flag), ctypes.byref(first), ctx.stream())
        print(f"nc={nc} n_coef={n_coef}: rc={rc} flag={flag.value} first={first.value}")
