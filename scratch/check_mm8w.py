"""k_mm8w (full-size entries on the matrix cores) through hb_matvec against exact Python integers."""
import ctypes, random, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, HbView, np_ptr
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = Context.get(P); lib = ctx.lib
rnd = random.Random(11)
def run(n_out, n_in, C, party_major_in, party_major_out, rows=False, edge=False):
    M = [[rnd.randrange(P) for _ in range(n_in)] for _ in range(n_out)]
    if edge:
        M[0] = [P - 1] * n_in; M[-1] = [0] * n_in; M[n_out // 2] = [1] * n_in
    X = [[rnd.randrange(P) for _ in range(n_in)] for _ in range(C)]
    if edge:
        X[0] = [P - 1] * n_in; X[1] = [0] * n_in; X[2] = [(1 << 256) - 1] * n_in; X[3] = [(1 << 255)] * n_in
    perm = list(range(n_in))
    if rows: rnd.shuffle(perm)
    h = ctypes.c_void_p()
    ctx.check(lib.hb_matrix_from_host(ctx.h, np_ptr(ctx.host_elems([v for r in M for v in r])), n_out, n_in, ctypes.byref(h), ctx.stream()), "from_host")
    # input buffer: element (c, l) in the chosen layout; values are NOT reduced for the edge rows (kernels reduce on entry)
    from honeybadgermpc_amd._capi import ints_to_limbs
    flat = [0] * (C * n_in)
    for c in range(C):
        for l in range(n_in):
            flat[(l * C + c) if party_major_in else (c * n_in + l)] = X[c][l]
    arr = np.zeros((C * n_in, 4), dtype=np.uint64)
    for i, v in enumerate(flat):
        for q in range(4): arr[i, q] = (v >> (64 * q)) & 0xFFFFFFFFFFFFFFFF
    xin = ctx.to_device(arr)
    out = ctx.empty(C * n_out)
    iv = HbView(1, C) if party_major_in else HbView(n_in, 1)
    ov = HbView(1, C) if party_major_out else HbView(n_out, 1)
    pr = np.array(perm, dtype=np.int32)
    ctx.check(lib.hb_matvec(ctx.h, h, ctx.ptr(xin), iv, np_ptr(pr) if rows else None, ctx.ptr(out), ov, C, ctx.stream()), "matvec")
    got = ctx.download_ints(out)
    bad = 0
    for c in range(C):
        for i in range(n_out):
            want = sum(M[i][l] * X[c][perm[l]] for l in range(n_in)) % P
            g = got[(i * C + c) if party_major_out else (c * n_out + i)]
            if g != want:
                bad += 1
                if bad < 4: print("MISMATCH", n_out, n_in, "c", c, "i", i, hex(g), hex(want))
    # check mode: expected = out; flip one element to see the flag
    mm = torch.zeros(1, dtype=torch.int32, device='cuda')
    cr = np.arange(n_out, dtype=np.int32)
    ctx.check(lib.hb_matvec_check(ctx.h, h, ctx.ptr(xin), iv, np_ptr(pr) if rows else None, ctx.ptr(out), ov, np_ptr(cr), n_out, ctx.ptr(mm), C, ctx.stream()), "check")
    ok_flag = int(mm.item())
    out2 = out.clone(); out2[(C * n_out) // 2, 0] ^= 1
    ctx.check(lib.hb_matvec_check(ctx.h, h, ctx.ptr(xin), iv, np_ptr(pr) if rows else None, ctx.ptr(out2), ov, np_ptr(cr), n_out, ctx.ptr(mm), C, ctx.stream()), "check")
    bad_flag = int(mm.item())
    print(f"n_out={n_out} n_in={n_in} C={C} pm_in={party_major_in} pm_out={party_major_out} rows={rows} edge={edge}: mismatches={bad} check_ok_flag={ok_flag} check_bad_flag={bad_flag}", flush=True)
    lib.hb_matrix_destroy(h)
    return bad == 0 and ok_flag == 0 and bad_flag == 1
ok = True
for args in [(16, 4, 256, False, False), (22, 22, 300, True, True, True, True), (64, 22, 259, False, True, False, True), (86, 86, 257, True, False, True),
             (100, 34, 256, False, False), (17, 9, 1000, True, True), (5, 5, 256, False, False), (48, 128, 256, True, True), (33, 40, 512, False, True, True)]:
    ok = run(*args) and ok
print("ALL OK" if ok else "FAILED")
# timing at the config-5 decode shape
n = 86; C = 6097
M = [[rnd.randrange(P) for _ in range(n)] for _ in range(n)]
h = ctypes.c_void_p()
ctx.check(lib.hb_matrix_from_host(ctx.h, np_ptr(ctx.host_elems([v for r in M for v in r])), n, n, ctypes.byref(h), ctx.stream()), "from_host")
g = torch.Generator(device='cuda'); g.manual_seed(1)
x = torch.randint(-(1 << 63), (1 << 63) - 1, (C * n, 4), dtype=torch.int64, device='cuda', generator=g); x[:, 3] &= (1 << 61) - 1
o = ctx.empty(C * n)
for shape, (nn, dd, CC) in {"cfg5 decode 86x86": (86, 86, 6097), "cfg3w decode 22x22": (22, 22, 47663), "cfg3w encode 64x22": (64, 22, 47663)}.items():
    M = [[rnd.randrange(P) for _ in range(dd)] for _ in range(nn)]
    h = ctypes.c_void_p()
    ctx.check(lib.hb_matrix_from_host(ctx.h, np_ptr(ctx.host_elems([v for r in M for v in r])), nn, dd, ctypes.byref(h), ctx.stream()), "from_host")
    x = torch.randint(-(1 << 63), (1 << 63) - 1, (CC * dd, 4), dtype=torch.int64, device='cuda', generator=g); x[:, 3] &= (1 << 61) - 1
    o = ctx.empty(CC * nn)
    for _ in range(3):
        ctx.check(lib.hb_matvec(ctx.h, h, ctx.ptr(x), HbView(1, CC), None, ctx.ptr(o), HbView(1, CC), CC, ctx.stream()), "mv")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        ctx.check(lib.hb_matvec(ctx.h, h, ctx.ptr(x), HbView(1, CC), None, ctx.ptr(o), HbView(1, CC), CC, ctx.stream()), "mv")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{shape}: {dt * 1e6:.1f} us per launch", flush=True)
