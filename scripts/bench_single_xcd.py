"""Round-3 experiment (EXPERIMENTS library): the single-read emit kernel with ONE LOOK-BACK CHAIN PER XCD (VERDICT r2 next-6 b).
Every XCD walks its own contiguous eighth of the tiles; its two look-backs (line index, column offsets) never leave the XCD.
What such a design would have to pay for on top -- the eighths' unknown starting offsets, i.e. columns laid out per eighth and
the batches across the seams repacked -- is GIVEN here: the chain starts are taken from the two-pass run of the same input.  So
the kernel writes the ordinary contiguous columns, its output can be compared bit for bit, and its time is a LOWER bound of
the variant.
    BLAZESEQ_HIP_LIB=blazeseq_amd/libblazeseq_hip_exp.so python scripts/bench_single_xcd.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import blazeseq_amd as B
from blazeseq_amd import _lib as L

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0, min_record_bytes=256)
ctx.set_option("timing_detail", 1)
n = ctx.generate_synthetic_device(reads, 150, 33, 73, "generic", 0, 0)
buf = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(reads, 150, 33, 73, "generic", buf.data_ptr(), buf.numel())
torch.cuda.synchronize()


def run(mode, steps=20):
    ctx.set_option("single_pass", mode)
    out = []
    for i in range(steps + 3):
        ctx.submit_device(buf.data_ptr(), n, 0, True)
        r = ctx.result()
        assert int(r.n_records) == reads and r.status == L.EOF, (mode, r.n_records, r.status, ctx.format_error())
        if i >= 3:
            out.append((r.ms_total, r.ms_aggregate, r.ms_emit, r.ms_rebase))
    a = np.array(out).mean(axis=0)
    return r, a


r2, t2 = run(0)
ref = (r2.seq().copy(), r2.qual().copy(), r2.id().copy(), r2.ends().copy(), r2.id_ends().copy(), r2.record_end().copy())
print(f"two passes              : kernels {t2[0]:.3f} ms (pass A {t2[1]:.3f}, emit {t2[2]:.3f}, rebase {t2[3]:.3f})", flush=True)
for mode, name in ((4, "one look-back chain per XCD, chain starts given"), (1, "one look-back chain over all tiles (round 1's k_fused<LB=true>)")):
    if mode == 4:
        run(0, steps=0)   # (the chain starts are read from the two-pass run's tile prefixes of this very input)
    r, t = run(mode)
    same = all(np.array_equal(x, y) for x, y in zip(ref, (r.seq(), r.qual(), r.id(), r.ends(), r.id_ends(), r.record_end())))
    print(f"{name:62s}: kernels {t[0]:.3f} ms (emit {t[2]:.3f}, rebase {t[3]:.3f})   output identical to two passes: {same}", flush=True)
    if mode == 4:
        run(0, steps=0)
