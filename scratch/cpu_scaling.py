import ctypes, time, sys, numpy as np
sys.path.insert(0,'.')
import oracle
P=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
n,d,C=64,22,47663
rng=np.random.default_rng(1)
polys=rng.integers(0,1<<62,size=(C*d,4),dtype=np.uint64)
x=oracle._limbs(list(range(1,n+1)),P); out=np.zeros((C*n,4),dtype=np.uint64)
lib=oracle.lib()
for th in (1,8,32,64,128,256):
    oracle.SetNumThreads(th)
    best=9
    for rep in range(3):
        t0=time.perf_counter()
        lib.orc_vandermonde_batch_evaluate(oracle._ptr(oracle._p(P)),oracle._ptr(x),n,oracle._ptr(polys),ctypes.c_long(C),d,oracle._ptr(out))
        best=min(best,time.perf_counter()-t0)
    print(th, round(best,4), round(C*n*d/best/1e6,1),"M mulmod/s", flush=True)
