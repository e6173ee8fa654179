"""How long the first half of a first-sight decode takes on the GPU (hb_quick_dec_arrivals: k_fs_build_z_cand, or the full-size builder), back to back on one stream."""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, np_ptr
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P); lib = ctx.lib
for n, t, omega in ((64, 21, False), (64, 21, True), (256, 85, True)):
    d = t + 1
    if omega:
        from honeybadgermpc_amd.field import GF
        from honeybadgermpc_amd.polynomial import EvalPoint
        pt = EvalPoint(GF(P), n, use_omega_powers=True)
        x = [pt(i).value for i in range(n)]
    else:
        x = list(range(1, n + 1))
    h = ctypes.c_void_p()
    assert lib.hb_quick_dec_create(ctx.h, np_ptr(ctx.host_elems(x)), n, ctypes.byref(h), ctx.stream()) == 0
    rng = np.random.Generator(np.random.PCG64(1))
    for n_coef in (d, 1):
        zs = [np.array(rng.permutation(n)[:d], dtype=np.int32) for _ in range(300)]
        for z in zs[:20]:
            assert lib.hb_quick_dec_arrivals(h, np_ptr(z), d, t, n_coef, ctx.stream()) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for z in zs[20:]:
            lib.hb_quick_dec_arrivals(h, np_ptr(z), d, t, n_coef, ctx.stream())
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n} t={t} omega={omega} n_coef={n_coef}: host enqueue {(t1-t0)/280*1e6:.1f} us, GPU (back to back) {(t2-t0)/280*1e6:.1f} us per first half", flush=True)
    lib.hb_quick_dec_destroy(h)
