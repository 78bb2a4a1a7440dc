"""DETERMINISTIC reproducer of round 5's "timing-dependent" mismatch: bzq_create zeroed the views pool's ticket with
hipMemset -- on the NULL stream -- while the ctx stream is a non-blocking stream: nothing orders that fill in front of the first
chunk's pass A.  Here the fill is made late ON PURPOSE: a spin kernel of a chosen length sits on the null stream when the ctx is
created (the fill queues behind it), and the first chunk -- 1.5 GB with pool tiles all along, so that pass A takes ~0.5 ms -- is
submitted at once.  When the spin ends inside pass A the ticket is reset under the tiles that are taking tickets: two tiles get the
same pool slot, one overwrites the other's entries, the join reads garbage.   BZQ_POOL_ZERO=0 is rounds 4-5's create (null stream),
BZQ_POOL_ZERO=1 (the default now) fills on the ctx stream.      python scripts/probes/pool_zero_race.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import blazeseq_amd as B
from oracle import oracle as O


def make_chunk(total_mb=1500):
    """ordinary 150 bp reads with a stretch of 8-byte records (2048 per tile: a pool tile) every ~520 KB: ~6 % of the tiles pooled"""
    normal = O.generate_synthetic(1600, 150, 150, 0, 40, "sanger").tobytes()        # ~500 KB
    tiny = b"@\nA\n+\n!\n" * 2600                                                    # ~20 KB: at least one whole tile of tiny records
    piece = normal + tiny
    reps = max(1, total_mb * 1_000_000 // len(piece))
    recs_piece = 1600 + 2600
    return np.frombuffer(piece * reps, dtype=np.uint8), reps * recs_piece


def run(pool_zero, spin_us, d_chunk, n, want_records, rate_hz):
    os.environ["BZQ_POOL_ZERO"] = str(pool_zero)
    torch.cuda.synchronize()
    if spin_us:
        torch.cuda._sleep(int(spin_us * 1e-6 * rate_hz))      # on torch's current stream = the NULL stream
    t0 = time.perf_counter()
    ctx = B.Context(B.ParserConfig(views_only=True), "generic", 4096, 0, min_record_bytes=8)
    t1 = time.perf_counter()
    ctx.submit_device(d_chunk.data_ptr(), n, 0, True)
    r = ctx.result()
    got = (int(r.n_records), int(r.status))
    ctx.close()
    return got, (t1 - t0) * 1e6


data, want = make_chunk()
d_chunk = torch.from_numpy(data.copy()).cuda()
n = data.size
# the spin kernel counts shader-clock cycles: calibrate cycles per second once
torch.cuda.synchronize(); t = time.perf_counter(); torch.cuda._sleep(200_000_000); torch.cuda.synchronize(); rate = 200_000_000 / (time.perf_counter() - t)
print(f"chunk {n / 1e6:.0f} MB, {want} records; spin counter {rate / 1e9:.3f} GHz", flush=True)
for pool_zero in (0, 1):
    wrong = []
    for spin_us in [0] + list(range(200, 4001, 100)):
        for rep in range(3):
            got, create_us = run(pool_zero, spin_us, d_chunk, n, want, rate)
            if got != (want, 6):
                wrong.append((spin_us, rep, got, round(create_us)))
    print(f"BZQ_POOL_ZERO={pool_zero} ({'hipMemset on the NULL stream: rounds 4-5' if pool_zero == 0 else 'hipMemsetAsync on the ctx stream: round 6'}): "
          f"{len(wrong)} wrong results of {3 * 40}; first: {wrong[:6]}", flush=True)
