"""Where a synchronous step's host time goes: wall clock of submit_device / result / bzq_batches, and the kernels' event time.
usage: python scripts/host_step_times.py [--views]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blazeseq_amd as B
from blazeseq_amd import _lib as L
views = "--views" in sys.argv
cfg = B.ParserConfig(views_only=views)
ctx = B.Context(cfg, "generic", 4096, 0)
n = ctx.generate_synthetic_device(10_000_000, 150, 33, 73, "generic", 0, 0, first=0, count=10_000_000, max_len=150)
buf = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(10_000_000, 150, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=0, count=10_000_000, max_len=150)
torch.cuda.synchronize()
arr = (L.BzqDeviceBatch * 2500)(); nout = C.c_uint64()
ts = [0.0, 0.0, 0.0]; ms = 0.0
K = 200
for it in range(K + 20):
    if it == 20: ctx.set_option('dump_host_times', 1)
    t0 = time.perf_counter(); ctx.submit_device(buf.data_ptr(), n, 0, True)
    t1 = time.perf_counter(); r = ctx.result()
    t2 = time.perf_counter()
    if not views: L.lib().bzq_batches(ctx.h, 4096, arr, 2500, C.byref(nout))
    t3 = time.perf_counter()
    if it >= 20:
        ts[0] += t1 - t0; ts[1] += t2 - t1; ts[2] += t3 - t2; ms += r.ms_total
ctx.set_option("dump_host_times", 1)
print(f"views={views} per step: submit {ts[0]/K*1e6:.1f} us, result (incl. wait) {ts[1]/K*1e6:.1f} us, batches {ts[2]/K*1e6:.1f} us, total {sum(ts)/K*1e6:.1f} us; kernels (events) {ms/K*1e3:.1f} us")
