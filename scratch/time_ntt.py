"""Timing of the LDS NTT on its own: hb_fft_batch_evaluate, C polynomials of d coefficients at the `order` powers of omega."""
import sys
import time

import torch

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context, np_ptr  # noqa: E402
from honeybadgermpc_amd.field import GF  # noqa: E402
from honeybadgermpc_amd.polynomial import EvalPoint  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    for n, d, c in ((64, 22, 47663), (16, 6, 65536), (256, 86, 6097)):
        ctx = Context.get(P)
        pt = EvalPoint(GF(P), n, use_omega_powers=True)
        om = ctx.host_elems([pt.omega.value])
        x = ctx.empty(c * d)
        x.random_(0, 1 << 62)
        out = ctx.empty(c * pt.order)

        def run():
            rc = ctx.lib.hb_fft_batch_evaluate(ctx.h, np_ptr(om), pt.order, ctx.ptr(x), c, d, pt.order, ctx.ptr(out), ctx.stream())
            assert rc == 0

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"NTT order {pt.order} d={d} C={c}: {dt * 1e6:.1f} us")


main()
