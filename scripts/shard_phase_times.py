"""Where the shard protocol's step time goes at world_size 1 (torchrun --nproc-per-node 1 scripts/shard_phase_times.py)."""
import os, sys, time
sys.path.insert(0, ".")
import torch, torch.distributed as dist
import blazeseq_amd as B
from blazeseq_amd import sharded as S
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)), device_id=torch.device("cuda", 0))
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
reads = 10_000_000
n = ctx.generate_synthetic_device(reads, 150, 33, 73, "generic", 0, 0)
shard = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(reads, 150, 33, 73, "generic", shard.data_ptr(), shard.numel())
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
for it in range(25):
    if it == 5: T.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    s = ctx.shard_scan(shard.data_ptr(), n); t = tick("shard_scan", t)
    local = [int(s.n_bytes), int(s.n_newlines), *[int(x) for x in s.first_nl], int(s.first_byte), int(s.last_byte)]
    summaries = S.gather_summaries(local, shard.device); t = tick("gather_summaries", t)
    plan = S.plan_shards(summaries)[0]; S.exchange_halo(shard, n, plan); t = tick("plan+halo", t)
    ctx.submit_shard(shard.data_ptr(), n, plan.halo_bytes, plan.lines_before, plan.prev_last_byte, 0, True); res = ctx.result(); t = tick("submit+result", t)
    S.gather_outcomes(int(res.n_records), int(res.seq_bytes), n, -1, shard.device); t = tick("gather_outcomes", t)
print({k: round(v / 20 * 1e3, 4) for k, v in T.items()}, "ms; sum", round(sum(T.values()) / 20 * 1e3, 4))
# the same box, the same process: one-shot parse, and the protocol without the per-phase synchronisations
for name, f in (("one-shot submit+result", lambda: (ctx.submit_device(shard.data_ptr(), n, 0, True), ctx.result())),
                ("parse_sharded", lambda: S.parse_sharded(ctx, shard, n, 0))):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 20 * 1e3, 4), "ms")
dist.destroy_process_group()
