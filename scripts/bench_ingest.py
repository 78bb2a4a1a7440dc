"""End-to-end (PCIe-inclusive) rate of the native ingest pipeline: file (page cache, /dev/shm) -> pinned -> device ->
parsed columns.  Never the bench `value` (that is HBM-resident); quoted in DESIGN.md.
   python scripts/bench_ingest.py [--reads 10000000] [--threads 8] [--chunk-mib 256]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import blazeseq_amd as B
from blazeseq_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--threads", type=int, nargs="+", default=[4, 8, 16])
ap.add_argument("--chunk-mib", type=int, nargs="+", default=[256])
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--direct", action="store_true", help="option ingest_direct: O_DIRECT reads (needs --dir on a real filesystem; tmpfs refuses O_DIRECT)")
ap.add_argument("--drop-caches", action="store_true", help="echo 3 > /proc/sys/vm/drop_caches before every run (cold file)")
args = ap.parse_args()

ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
if args.direct:
    ctx.set_option("ingest_direct", 1)
rec = ctx.generate_synthetic_device(args.reads, 150, 33, 73, "generic", count=1)
n = rec * args.reads
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(args.reads, 150, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=0, count=args.reads)
torch.cuda.synchronize()
host = buf[:n].cpu().numpy()
path = os.path.join(args.dir if os.path.isdir(args.dir) else "/tmp", "bzq_ingest_bench.fastq")
host.tofile(path)
del buf
print(f"file {path}: {n/1e9:.2f} GB, host cores {os.cpu_count()}", flush=True)
# plain H2D of the same bytes from pinned memory, for reference
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pin.numpy()[:] = host
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"plain pinned H2D copy: {n/dt/1e9:.1f} GB/s")
del pin, dev
for chunk in args.chunk_mib:
    for th in args.threads:
        best = None
        for rep in range(3):
            if args.drop_caches:
                os.sync()
                try:
                    open("/proc/sys/vm/drop_caches", "w").write("3\n")
                except OSError as e:
                    print("drop_caches:", e)
            t0 = time.perf_counter()
            ing = B.Ingest(ctx, path, chunk_bytes=chunk << 20, n_threads=th)
            t1 = time.perf_counter()
            taken, total = 0, 0
            while True:
                res = ing.next(taken)
                taken = int(res.n_records)
                total += taken
                if int(res.status) != L.OK:
                    break
            dt = time.perf_counter() - t1
            st = ing.stats()
            ing.close()
            assert total == args.reads, (total, args.reads)
            r = (n / dt / 1e9, dt, st.read_s, st.wait_s, t1 - t0)
            if best is None or r[0] > best[0]: best = r
        print(f"[direct_io={int(st.direct_io)} numa_node={int(st.numa_node)}] chunk {chunk} MiB, {th} reader threads: {best[0]:.1f} GB/s end to end ({best[1]*1e3:.0f} ms; reader busy {best[2]*1e3:.0f} ms, "
              f"consumer waiting {best[3]*1e3:.0f} ms, open {best[4]*1e3:.0f} ms)", flush=True)
os.remove(path)

# ---- compressed input: gzip stream (zlib gzread, serial) and BGZF (block-parallel inflate) ---------------------------
if os.environ.get("BZQ_BENCH_COMPRESSED"):
    import struct, zlib, gzip
    reads = 3_000_000
    ctx2 = B.Context(B.ParserConfig(), "generic", 4096, 0)
    rec = ctx2.generate_synthetic_device(reads, 150, 33, 73, "generic", count=1)
    n2 = rec * reads
    buf = torch.empty(n2 + 64, dtype=torch.uint8, device="cuda")
    ctx2.generate_synthetic_device(reads, 150, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=0, count=reads)
    raw = buf[:n2].cpu().numpy().tobytes()
    del buf
    t0 = time.perf_counter()
    gz_path = os.path.join(args.dir, "bzq_ingest_bench.fastq.gz")
    with open(gz_path, "wb") as f:
        f.write(gzip.compress(raw, 1))
    blocks = []
    for i in list(range(0, len(raw), 65280)) + [None]:
        chunk = b"" if i is None else raw[i:i + 65280]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        blocks.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1)
                      + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    bg_path = os.path.join(args.dir, "bzq_ingest_bench.fastq.bgz")
    with open(bg_path, "wb") as f:
        f.write(b"".join(blocks))
    print(f"compressed {n2/1e9:.2f} GB twice in {time.perf_counter()-t0:.1f} s: gzip {os.path.getsize(gz_path)/1e9:.2f} GB, BGZF {os.path.getsize(bg_path)/1e9:.2f} GB", flush=True)
    for name, path, threads in (("gzip (one zlib stream)", gz_path, [1]), ("BGZF", bg_path, [1, 8, 32, 64])):
        for th in threads:
            t1 = time.perf_counter()
            ing = B.Ingest(ctx2, path, chunk_bytes=64 << 20, n_threads=th)
            taken, total = 0, 0
            while True:
                res = ing.next(taken)
                taken = int(res.n_records); total += taken
                if int(res.status) != L.OK: break
            dt = time.perf_counter() - t1
            ing.close()
            assert total == reads
            print(f"{name}, {th} inflate threads: {n2/dt/1e9:.2f} GB/s of FASTQ ({dt*1e3:.0f} ms)", flush=True)
    os.remove(gz_path); os.remove(bg_path)
