"""File -> device -> FASTA records through bzq_fasta_ingest_* (page-cache file, plain): end-to-end GB/s.
python scripts/bench_fasta_ingest.py [threads] [chunk_MiB]"""
import json
import os
import sys
import time

sys.path.insert(0, ".")
import blazeseq_amd as B

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chunk = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
ctx = B.FastaContext()
t = ctx.generate_synthetic_device(1_500_000, 200, 3800, 60)
path = "/dev/shm/bzq_bench.fasta" if os.path.isdir("/dev/shm") else "/tmp/bzq_bench.fasta"
t.cpu().numpy().tofile(path)
n = t.numel()
del t
for rep in range(3):
    ing = B.FastaIngest(ctx, path, chunk, threads)
    t0 = time.perf_counter()
    recs = 0
    while True:
        res = ing.next()
        recs += int(res.n_records)
        if int(res.status) != 0:
            break
    dt = time.perf_counter() - t0
    st = ing.stats()
    ing.close()
    assert int(res.status) == 6 and recs == 1_500_000, (res.status, recs)
print(json.dumps({"workload": "FASTA file -> device -> records", "bytes": n, "records": recs, "threads": threads, "chunk_MiB": chunk >> 20,
                  "seconds": round(dt, 4), "GB_per_s": round(n / dt / 1e9, 2), "chunks": int(st.chunks)}))
os.remove(path)
