"""bench.py's ingest_mode (file -> records, wall clock) over chunk sizes and reader-thread counts.
usage: python scripts/sweep_ingest.py [mode ...] [CHUNK_MIB:THREADS ...]   (modes: plain bgzf gzip)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import blazeseq_amd as B

modes = tuple(a for a in sys.argv[1:] if ":" not in a) or ("plain", "bgzf", "gzip")
configs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:] if ":" in a] or [(256, 8), (512, 8), (1024, 8), (256, 16), (1024, 16), (128, 8)]
dev = torch.device("cuda:0")
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0, min_record_bytes=256)
n = 200_000
nb = ctx.generate_synthetic_device(10_000_000, 150, 33, 73, "generic", count=n)
shard = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
ctx.generate_synthetic_device(10_000_000, 150, 33, 73, "generic", d_out=shard.data_ptr(), cap=shard.numel(), count=n)
assert nb == n * 318, nb
torch.cuda.synchronize()
for chunk, threads in configs:
    r = bench.ingest_mode(shard[:nb], 318, dev, 0, threads=threads, chunk_mib=chunk, modes=modes)
    print(f"chunk {chunk} MiB, {threads} threads: " + ", ".join(f"{m} {r[m]['value_warm_cache']} GB/s warm ({r[m]['warm_cache']['ms']} ms; first file {r[m]['value']})" for m in modes), flush=True)
