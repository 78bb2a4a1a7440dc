"""Does `import torch` return when the library has already used the GPU in this process?  (blazeseq_amd/_lib.py loads torch's copy of
libamdhip64 first so that both share one runtime.)  Prints the seconds each step took."""
import sys, time
t0 = time.perf_counter()
import blazeseq_amd as B
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
data = b"@r1\nACGT\n+\nIIII\n" * 1000
res = ctx.parse(data) if hasattr(ctx, "parse") else None
t1 = time.perf_counter()
print(f"library used the GPU: {t1 - t0:.1f} s", flush=True)
import torch
t2 = time.perf_counter()
print(f"import torch afterwards: {t2 - t1:.1f} s; cuda available: {torch.cuda.is_available()}", flush=True)
x = torch.zeros(16, device="cuda"); torch.cuda.synchronize()
print(f"torch allocated on the same runtime: {time.perf_counter() - t2:.1f} s", flush=True)
