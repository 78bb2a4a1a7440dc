"""GPU BGZF inflate rate: a slice of the benchmark's synthetic 150 bp FASTQ is compressed with zlib (level 6, 65280-byte
blocks = bgzip), the compressed blocks are repeated on the device to --gb of OUTPUT, one call inflates them all.
    python scripts/bench_bgzf_inflate.py [--gb 2] [--slice-mb 48] [--level 6]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch   # (before the library: INTEGRATION.md)
import blazeseq_amd as B
from blazeseq_amd import _lib as L
from tests.bgzf_util import bgzf_compress

ap = argparse.ArgumentParser()
ap.add_argument("--gb", type=float, default=2.0)
ap.add_argument("--slice-mb", type=int, default=48)
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--ms", type=int, default=0, help="1: eight blocks per wave (bzq_inflate_ms.hpp, the round-4 experiment); 0: one block per wave (the product)")
args = ap.parse_args()
ctx = B.Context()
if args.ms:
    ctx.set_option("inflate_ms", 1)   # (EXPERIMENTS library only: BLAZESEQ_HIP_LIB=blazeseq_amd/libblazeseq_hip_exp.so)
lib = L.lib()
n_rec = args.slice_mb * (1 << 20) // 318
size = ctx.generate_synthetic_device(n_rec, 150, 33, 73, "generic", 0, 0)
buf = torch.empty(size + 64, dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(n_rec, 150, 33, 73, "generic", buf.data_ptr(), buf.numel())
torch.cuda.synchronize()
plain = buf[:size].cpu().numpy().tobytes()
t0 = time.perf_counter()
comp = np.frombuffer(bgzf_compress(plain, level=args.level, eof_marker=False), dtype=np.uint8)
t_host_c = time.perf_counter() - t0
blocks, n, consumed, out_bytes = ctx.bgzf_scan(comp)
assert consumed == comp.size and out_bytes == len(plain)
# zlib on this host, one thread, the same blocks (what one reader thread of the host path does)
import zlib
t0 = time.perf_counter()
k = 0
for i in range(min(n, 200)):
    b = blocks[i]
    k += len(zlib.decompress(comp[b.comp_offset + 18:b.comp_offset + b.comp_size - 8].tobytes(), -15))
t_host = (time.perf_counter() - t0) / max(k, 1)
reps = max(1, int(args.gb * 1e9 / out_bytes))
tab = (L.BzqBgzfBlock * (n * reps))()
for r in range(reps):
    for i in range(n):
        s, d = blocks[i], tab[r * n + i]
        d.comp_offset, d.comp_size, d.out_size, d.crc32, d.out_offset = s.comp_offset, s.comp_size, s.out_size, s.crc32, s.out_offset + r * out_bytes
d_comp = torch.from_numpy(comp.copy()).cuda()
d_out = torch.empty(out_bytes * reps + 64, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def run():
    try:
        ctx.bgzf_inflate(d_comp.data_ptr(), comp.size, tab, n * reps, d_out.data_ptr(), out_bytes * reps)
    except RuntimeError:
        if not os.environ.get("BZQ_IGNORE_FAIL"):   # (timing experiments with builds that decode wrongly on purpose)
            raise
for _ in range(2):
    run()
t0 = time.perf_counter()
for _ in range(args.steps):
    run()
dt = (time.perf_counter() - t0) / args.steps
ok = bool((d_out[:out_bytes] == buf[:size]).all()) and bool((d_out[out_bytes * (reps - 1):out_bytes * reps] == buf[:size]).all())
print(f"BGZF level {args.level}: {len(plain)/1e6:.0f} MB of FASTQ -> {comp.size/1e6:.0f} MB ({len(plain)/comp.size:.2f}x), {n} blocks; "
      f"x{reps} = {out_bytes*reps/1e9:.2f} GB out, {n*reps} blocks")
print(f"GPU inflate  {dt*1e3:8.2f} ms  {out_bytes*reps/dt/1e9:7.1f} GB/s of FASTQ ({comp.size*reps/dt/1e9:.1f} GB/s compressed)  identical={ok}")
print(f"host zlib    one thread {1/t_host/1e9:6.3f} GB/s of FASTQ")
