"""Wall-clock time of k_mm8w launches (config 3's fused decodes, the 64 x 22 encode, config 5's 86 x 86 decode) for the library named
by HBMPC_HIP_LIB; with several libraries on the command line it runs itself once per library.  Results are not checked here
(variants built with gen_mm8w.py's ablation knobs compute nonsense): tests do that."""
import os, subprocess, sys
if len(sys.argv) > 1:
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["HBMPC_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=300)
        print(f"{os.path.basename(lib):40s} {r.stdout.strip()}" + (f"  [stderr: {r.stderr.strip()[-200:]}]" if r.returncode else ""), flush=True)
    sys.exit(0)
import ctypes, random, time
import torch
sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context, HbView, np_ptr
from honeybadgermpc_amd.device import BatchOpen
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P); lib = ctx.lib
rnd = random.Random(3)
g = torch.Generator(device='cuda'); g.manual_seed(1)
def timed(run, reps=30):
    for _ in range(5): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6
out = []
n_, t_ = 64, 21
d_ = t_ + 1
B_ = 1 << 20
order = list(range(n_)); random.Random(7).shuffle(order)
op = BatchOpen(P, n_, t_, z=order[:d_], zc=order[d_:d_ + t_], max_shares=B_)
sh = torch.randint(0, 1 << 62, (B_, 4), dtype=torch.int64, device='cuda', generator=g)
cols = op.r1_encode(sh)
out.append("R1 %.1f" % timed(lambda: op.r1_decode(cols, B_)))
out.append("R2 %.1f" % timed(lambda: op.r2_decode(cols, B_)))
for nn, dd, CC in [(64, 22, 47663), (86, 86, 6097), (171, 86, 6097), (256, 86, 6097)]:
    M = [[rnd.randrange(P) for _ in range(dd)] for _ in range(nn)]
    h = ctypes.c_void_p()
    ctx.check(lib.hb_matrix_from_host(ctx.h, np_ptr(ctx.host_elems([v for r in M for v in r])), nn, dd, ctypes.byref(h), ctx.stream()), "from_host")
    x = torch.randint(-(1 << 63), (1 << 63) - 1, (CC * dd, 4), dtype=torch.int64, device='cuda', generator=g); x[:, 3] &= (1 << 61) - 1
    o = ctx.empty(CC * nn)
    out.append("%dx%d %.1f" % (nn, dd, timed(lambda: ctx.check(lib.hb_matvec(ctx.h, h, ctx.ptr(x), HbView(1, CC), None, ctx.ptr(o), HbView(1, CC), CC, ctx.stream()), "mv"))))
print("  ".join(out) + " us")
