"""RCCL sanity on a 1-GPU box: a one-rank nccl process group (init, barrier, all_reduce, all_gather_into_tensor, batch_isend_irecv to self is not
possible) -- what bench.py's multi-rank path needs from the runtime, minus the peers."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
out = torch.empty(8, 4, dtype=torch.int64, device="cuda"); src = torch.arange(32, dtype=torch.int64, device="cuda").reshape(8, 4)
dist.all_gather_into_tensor(out, src)
torch.cuda.synchronize()
print("rccl ok", torch.cuda.nccl.version(), dist.get_world_size(), bool(torch.equal(out, src)))
dist.destroy_process_group()
