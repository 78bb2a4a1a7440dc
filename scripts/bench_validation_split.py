"""Where config 3's 6 % go: the headline input through batch mode with no validation, check_ascii only, check_quality only, both.
usage: python scripts/bench_validation_split.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blazeseq_amd as B
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reads = 10_000_000
ctx0 = B.Context(B.ParserConfig(), "generic", 4096, 0, min_record_bytes=256)
n = ctx0.generate_synthetic_device(reads, 150, 33, 73, "generic", count=reads)
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
ctx0.generate_synthetic_device(reads, 150, 33, 73, "generic", d_out=buf.data_ptr(), cap=buf.numel(), count=reads)
torch.cuda.synchronize()
for name, ca, cq in (("none", False, False), ("ascii", True, False), ("quality", False, True), ("both", True, True), ("none", False, False)):
    c = B.Context(B.ParserConfig(check_ascii=ca, check_quality=cq, quality_schema="sanger" if cq else None), "generic", 4096, 0, min_record_bytes=256)
    c.set_option("timing_detail", 1)
    for _ in range(100):
        c.submit_device(buf.data_ptr(), n, 0, True); r = c.result()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); me = ma = 0.0
    for _ in range(steps):
        c.submit_device(buf.data_ptr(), n, 0, True); r = c.result(); me += r.ms_emit; ma += r.ms_aggregate
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:8s} step {dt * 1e3:.4f} ms   emit {me / steps:.4f}   pass A {ma / steps:.4f}", flush=True)
    c.close()
