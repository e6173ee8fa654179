import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
from honeybadgermpc_amd.field import GF
from honeybadgermpc_amd.polynomial import EvalPoint
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P); lib = ctx.lib
n, t = int(sys.argv[1]), int(sys.argv[2]); d = t + 1
pt = EvalPoint(GF(P), n, use_omega_powers=True)
x = [pt(i).value for i in range(n)]
h = ctypes.c_void_p()
assert lib.hb_quick_dec_create(ctx.h, np_ptr(ctx.host_elems(x)), n, ctypes.byref(h), ctx.stream()) == 0
rng = np.random.Generator(np.random.PCG64(1))
for n_coef in (d, 1):
    zs = [np.array(rng.permutation(n)[:d], dtype=np.int32) for _ in range(60)]
    for z in zs:
        assert lib.hb_quick_dec_arrivals(h, np_ptr(z), d, t, n_coef, ctx.stream()) == 0
        torch.cuda.synchronize()
