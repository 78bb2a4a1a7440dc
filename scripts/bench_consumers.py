"""Rates of the device-side consumers (bzq_consumers.hpp) on the bench workload, columns resident in HBM.
   python scripts/bench_consumers.py [--reads 10000000]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blazeseq_amd as B

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
args = ap.parse_args()
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
rec = ctx.generate_synthetic_device(args.reads, 150, 33, 73, "generic", count=1)
n = rec * args.reads
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(args.reads, 150, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=0, count=args.reads)
ctx.submit_device(buf.data_ptr(), n, 0, True)
res = ctx.result()
d = B.DeviceFastqBatch(ctx, ctx.batch_view(0, int(res.n_records)))   # the whole chunk as one device batch
R, S = d.num_records, d.seq_len
scores = torch.empty(R, dtype=torch.int32, device="cuda")
sums = torch.empty(R, dtype=torch.int64, device="cuda")
REF = b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"   # REF_40BP, examples/nw_gpu/execution.mojo:36

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

t = timed(lambda: d.nw_scores(REF, scores.data_ptr()), 3)
print(f"nw_scores      {R} reads x 150 bp vs 40 bp reference: {t*1e3:.1f} ms = {R/t/1e6:.0f} M alignments/s, {R*150*40/t/1e12:.2f} T cell updates/s")
t = timed(lambda: d.quality_sums(sums.data_ptr()))
print(f"quality_sums   {t*1e3:.2f} ms = {S/t/1e9:.0f} GB/s of quality bytes")
t = timed(lambda: d.histogram('sequence'))
print(f"histogram(seq) {t*1e3:.2f} ms = {S/t/1e9:.0f} GB/s")
import ctypes as C
from blazeseq_amd import _lib as L
res._cumulative()
t = timed(lambda: L.lib().bzq_column_gc_counts(ctx.h, C.c_void_p(res.d_seq), C.c_void_p(res.d_ends), R, S, C.c_void_p(sums.data_ptr())))
print(f"gc_counts      {t*1e3:.2f} ms = {S/t/1e9:.0f} GB/s of sequence bytes")
