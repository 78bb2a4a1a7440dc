"""What this box's HBM delivers to plain streaming kernels (torch's own copy / reduce), for reading roofline.frac:
   python scripts/hbm_probe.py [bytes]"""
import sys, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_180_000_000
n -= n % 16
a = torch.empty(n // 4, dtype=torch.int32, device="cuda").random_()
b = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: b.copy_(a)); print(f"copy   {n/1e9:.2f} GB: {ms:.3f} ms  {2*n/ms/1e6:.0f} GB/s (read+write)")
ms = t(lambda: a.sum());    print(f"sum    {n/1e9:.2f} GB: {ms:.3f} ms  {n/ms/1e6:.0f} GB/s (read)")
ms = t(lambda: b.fill_(7)); print(f"fill   {n/1e9:.2f} GB: {ms:.3f} ms  {n/ms/1e6:.0f} GB/s (write)")
ms = t(lambda: torch.bitwise_xor(a, 5, out=b)); print(f"xor    {n/1e9:.2f} GB: {ms:.3f} ms  {2*n/ms/1e6:.0f} GB/s (read+write)")
