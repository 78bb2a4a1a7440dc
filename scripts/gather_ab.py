"""A/B of the tiny all_gather used by the shard protocol (torchrun --nproc-per-node 1 scripts/gather_ab.py)."""
import os, sys, time
sys.path.insert(0, ".")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29656")
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)), device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
world = dist.get_world_size()
row = list(range(10))

def a():
    mine = torch.tensor(row, dtype=torch.int64, device=dev)
    out = torch.empty(world * len(row), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().view(world, len(row)).tolist()

h_in = torch.empty(len(row), dtype=torch.int64, pin_memory=True); d_in = torch.empty(len(row), dtype=torch.int64, device=dev)
d_out = torch.empty(world * len(row), dtype=torch.int64, device=dev); h_out = torch.empty(world * len(row), dtype=torch.int64, pin_memory=True)
import numpy as np
h_in_np = h_in.numpy(); h_out_np = h_out.numpy()
def b():
    h_in_np[:] = row
    d_in.copy_(h_in, non_blocking=True)
    dist.all_gather_into_tensor(d_out, d_in)
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return h_out_np.reshape(world, len(row)).tolist()

g = dist.new_group(backend="gloo")
def c():
    out = [None] * world
    dist.all_gather_object(out, row, group=g)
    return out
t_in = torch.tensor(row, dtype=torch.int64); t_out = torch.empty(world * len(row), dtype=torch.int64)
def d():
    t_in[:] = torch.tensor(row)
    dist.all_gather_into_tensor(t_out, t_in, group=g)
    return t_out.view(world, len(row)).tolist()

for name, f in (("tensor+cpu()", a), ("pinned staging", b), ("gloo object", c), ("gloo tensor", d)):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): f()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 300 * 1e6, 1), "us")
dist.destroy_process_group()
