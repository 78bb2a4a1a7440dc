"""End to end from a BGZF file (page cache): reader threads -> pinned -> PCIe -> [inflate on the device | on the reader
threads] -> parser.  The file is a zlib-compressed slice of the benchmark's synthetic 150 bp FASTQ repeated to --gb of FASTQ
(BGZF blocks are independent, the slice ends at a record end, so the repetition is a valid file).
    python scripts/bench_ingest_bgzf.py [--gb 3] [--level 6] [--dir /dev/shm]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blazeseq_amd as B
from blazeseq_amd import _lib as L
from tests.bgzf_util import bgzf_compress

ap = argparse.ArgumentParser()
ap.add_argument("--gb", type=float, default=3.0)
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--slice-mb", type=int, default=48)
ap.add_argument("--chunk-mib", type=int, default=256)
ap.add_argument("--dir", default="/dev/shm")
args = ap.parse_args()
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
n_rec = args.slice_mb * (1 << 20) // 318
size = ctx.generate_synthetic_device(n_rec, 150, 33, 73, "generic", 0, 0)
buf = torch.empty(size + 64, dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(n_rec, 150, 33, 73, "generic", buf.data_ptr(), buf.numel())
torch.cuda.synchronize()
plain = buf[:size].cpu().numpy().tobytes()
comp = bgzf_compress(plain, level=args.level, eof_marker=False)
reps = max(1, int(args.gb * 1e9 / len(plain)))
path = os.path.join(args.dir if os.path.isdir(args.dir) else "/tmp", "bzq_ingest_bench.fastq.bgz")
with open(path, "wb") as f:
    for _ in range(reps):
        f.write(comp)
    f.write(bgzf_compress(b"", eof_marker=True))
total_plain, total_rec = len(plain) * reps, n_rec * reps
print(f"{path}: {os.path.getsize(path)/1e9:.2f} GB compressed (level {args.level}, {len(plain)/len(comp):.2f}x) = {total_plain/1e9:.2f} GB of FASTQ, {total_rec} records", flush=True)
for gpu, threads in ((1, 8), (1, 16), (0, 8), (0, 32), (0, 64)):
    c = B.Context(B.ParserConfig(), "generic", 4096, 0)
    c.set_option("ingest_gpu_inflate", gpu)
    best = None
    for rep in range(2):
        t1 = time.perf_counter()
        ing = B.Ingest(c, path, chunk_bytes=args.chunk_mib << 20, n_threads=threads)
        taken, total = 0, 0
        while True:
            res = ing.next(taken)
            taken = int(res.n_records); total += taken
            if int(res.status) != L.OK:
                break
        dt = time.perf_counter() - t1
        st = ing.stats()
        ing.close()
        assert total == total_rec, (total, total_rec)
        if best is None or dt < best[0]:
            best = (dt, st.read_s, st.wait_s)
    print(f"inflate on the {'device' if gpu else 'reader threads'}, {threads} reader threads: {total_plain/best[0]/1e9:6.2f} GB/s of FASTQ end to end "
          f"({best[0]*1e3:.0f} ms; reader busy {best[1]*1e3:.0f} ms, consumer waiting {best[2]*1e3:.0f} ms)", flush=True)
os.remove(path)
