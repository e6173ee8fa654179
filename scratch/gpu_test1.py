import ctypes, random, time, sys
import numpy as np, torch
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context, HbView, np_ptr, ints_to_limbs, limbs_to_ints
import oracle
P=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
def run(p, n, d, C, seed=0):
    random.seed(seed)
    ctx=Context.get(p); lib=ctx.lib
    x=[i+1 for i in range(n)]
    polys=[[random.randrange(p) for _ in range(d)] for _ in range(C)]
    want=oracle.vandermonde_batch_evaluate(x,polys,p)
    din=ctx.upload_ints([v for r in polys for v in r]); dout=ctx.empty(C*n)
    rc=lib.hb_vandermonde_batch_evaluate(ctx.h,np_ptr(ctx.host_elems(x)),n,ctx.ptr(din),C,d,ctx.ptr(dout),ctx.stream()); ctx.check(rc,"eval")
    got=ctx.download_ints(dout); got=[got[i*n:(i+1)*n] for i in range(C)]
    ok1 = got==want
    # interpolate from first d points
    ys=[r[:d] for r in want]
    din2=ctx.upload_ints([v for r in ys for v in r]); dout2=ctx.empty(C*d)
    rc=lib.hb_vandermonde_batch_interpolate(ctx.h,np_ptr(ctx.host_elems(x[:d])),d,ctx.ptr(din2),C,ctx.ptr(dout2),ctx.stream()); ctx.check(rc,"interp")
    got2=ctx.download_ints(dout2); got2=[got2[i*d:(i+1)*d] for i in range(C)]
    ok2 = got2==polys
    print(f"p={hex(p)[:12]} n={n} d={d} C={C}: eval {'OK' if ok1 else 'FAIL'} interp {'OK' if ok2 else 'FAIL'}", flush=True)
    if not ok1:
        for c in range(C):
            for i in range(n):
                if got[c][i]!=want[c][i]: print(" first diff", c,i,got[c][i],want[c][i]); break
            else: continue
            break
    return ok1 and ok2
allok=True
for (p,n,d,C) in [(P,4,2,3),(P,4,2,256),(P,16,6,100),(P,64,22,130),(13,4,2,10),(53,22,8,70),(P,7,3,65),(P,100,34,5),((1<<256)-189,10,4,64),(P,33,33,64),(P,1,1,1)]:
    allok &= run(p,n,d,C)
# singular
ctx=Context.get(P); lib=ctx.lib
dd=ctx.empty(4); rc=lib.hb_vandermonde_batch_interpolate(ctx.h,np_ptr(ctx.host_elems([1,1])),2,ctx.ptr(dd),2,ctx.ptr(dd),ctx.stream()); print("singular rc",rc)
print("ALL OK" if allok else "SOME FAIL")
# timing cfg3
n,d,B=64,22,1<<20; C=(B+d-1)//d
g=torch.Generator(device='cuda'); g.manual_seed(1)
din=torch.randint(0,2**62,(C*d,4),dtype=torch.int64,device='cuda',generator=g); din[:,3]&=(1<<61)-1   # < 2^253 < p
dout=ctx.empty(C*n)
x=ctx.host_elems([i+1 for i in range(n)])
for layout in ["CN","NC"]:
    V=ctypes.c_void_p(); rc=lib.hb_vand_matrix_create(ctx.h,np_ptr(x),n,d,ctypes.byref(V),ctx.stream()); ctx.check(rc,"V")
    iv=HbView(d,1); ov=HbView(n,1) if layout=="CN" else HbView(1,C)
    for it in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for r in range(5):
            rc=lib.hb_matvec(ctx.h,V,ctx.ptr(din),iv,None,ctx.ptr(dout),ov,C,ctx.stream())
        torch.cuda.synchronize(); t=(time.perf_counter()-t0)/5
    prods=C*n*d
    print(f"encode n=64 d=22 C={C} out={layout}: {t*1e3:.3f} ms  {prods/t/1e9:.1f} G prod-acc/s  MAD rate {prods*81/t/1e12:.2f} T/s")
# decode timing
xi=ctx.host_elems([i+1 for i in range(d)])
Vi=ctypes.c_void_p(); rc=lib.hb_vand_inverse_create(ctx.h,np_ptr(xi),d,ctypes.byref(Vi),ctx.stream()); ctx.check(rc,"Vi")
dout2=ctx.empty(C*d)
for iv,ov,name in [(HbView(d,1),HbView(d,1),"CN->CN"),(HbView(1,C),HbView(1,C),"NC->NC")]:
    for it in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for r in range(5):
            rc=lib.hb_matvec(ctx.h,Vi,ctx.ptr(din),iv,None,ctx.ptr(dout2),ov,C,ctx.stream())
        torch.cuda.synchronize(); t=(time.perf_counter()-t0)/5
    prods=C*d*d
    print(f"decode d=22 C={C} {name}: {t*1e3:.3f} ms  {prods/t/1e9:.1f} G prod-acc/s  MAD rate {prods*81/t/1e12:.2f} T/s")
