"""HBM rates of plain torch kernels on this box: fill, copy, read (sum), at 256 MB and 1 GB."""
import time, torch
for mb in (97, 256, 1024):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
    a.fill_(3); b.fill_(1); torch.cuda.synchronize()
    def t(f, reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    tf = t(lambda: a.fill_(5)); tc = t(lambda: b.copy_(a)); ts = t(lambda: a.sum())
    print(f"{mb:5d} MB: fill {mb / 1024 / tf / 1.024:6.2f} TB/s ({tf * 1e6:6.1f} us)   copy {2 * mb / 1024 / tc / 1.024:6.2f} TB/s r+w ({tc * 1e6:6.1f} us)   read {mb / 1024 / ts / 1.024:6.2f} TB/s ({ts * 1e6:6.1f} us)")
