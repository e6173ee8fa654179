import time, sys
sys.path.insert(0, ".")
import torch
from honeybadgermpc_amd.device import BatchOpen
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
for (n, t) in [(64, 21), (16, 5), (64, 21), (40, 13)]:
    t0 = time.perf_counter()
    op = BatchOpen(P, n, t, z=list(range(t + 1)), zc=list(range(t + 1, 2 * t + 1)), max_shares=1000)
    torch.cuda.synchronize()
    print(n, t, "plan create", round(time.perf_counter() - t0, 3), "s", op.uses_matrix_cores())
