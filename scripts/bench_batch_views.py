"""Host cost of handing out the batches of one parsed chunk: `for k in range(n_batches): bzq_batch_view(k * 4096, 4096)` --
the loop behind `for batch in parser.batches(4096): batch.to_device()` once the chunk is parsed (the reference pays 10
allocations, 10 copies and 3 synchronisations per batch there, record_batch.mojo:308-411)."""
import sys, time
sys.path.insert(0, ".")
import torch
import blazeseq_amd as B

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0, min_record_bytes=256)
nb = ctx.generate_synthetic_device(reads, 150, 33, 73, "generic")
t = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(reads, 150, 33, 73, "generic", t.data_ptr(), t.numel())
for rep in range(3):
    ctx.submit_device(t.data_ptr(), nb, 0, True)
    res = ctx.result()
    n = int(res.n_records)
    t0 = time.perf_counter()
    k = 0
    for first in range(0, n, 4096):
        v = ctx.batch_view(first, 4096)
        k += 1
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for first in range(7, n, 4096 * 64):      # unaligned views: a small kernel + a synchronisation each
        v = ctx.batch_view(first, 4096)
    du = (time.perf_counter() - t1) / len(range(7, n, 4096 * 64))
print(f"{k} aligned batch views of one {nb / 1e9:.2f} GB chunk: {dt * 1e3:.2f} ms = {dt / k * 1e6:.2f} us per view "
      f"({nb / dt / 1e9:.0f} GB/s of FASTQ handed out); an unaligned view: {du * 1e6:.1f} us")
