"""-m gpu: BASELINE config 5's shard shape at FULL size on one GPU -- one rank's byte range of the 625 M-read, 200 GB
synthetic file (9-digit headers -> 320 B/record, 25 GB per shard), cut at a byte that is NOT a record start, through
bzq_shard_scan -> bzq_submit_shard with the neighbours' contribution computed analytically (the line index of the
shard's first byte, the byte before it, the halo of the next shard's head).  Size-independent checks, no oracle run: record
counts, first / last owned record, every per-record array as a closed form, all three columns as slices of the input
viewed as a [records, 320] matrix (the method of tests/test_gpu_fullsize.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOTAL_READS = 625_000_000
REC = 320                      # "@read_%09d\n" 16 + 150 + "\n+\n" 3 + 150 + "\n" 1
WORLD = 8
SHIFT = 144                    # interior cuts sit 144 bytes into a record (inside its sequence line; keeps 16-byte alignment)


def shard_range(rank):
    total = TOTAL_READS * REC
    lo = total * rank // WORLD + (SHIFT if rank else 0)
    hi = total * (rank + 1) // WORLD + (SHIFT if rank + 1 < WORLD else 0)
    return lo, hi


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_config5_shard_of_the_200gb_file(rank):
    import torch
    import blazeseq_amd as B
    from tests.test_gpu_fullsize import _view
    free, _ = torch.cuda.mem_get_info()
    if free < 130 << 30:
        pytest.skip("needs ~130 GB of free HBM")
    lo, hi = shard_range(rank)
    n = hi - lo
    assert lo % REC == (SHIFT if rank else 0) and (WORLD == rank + 1 or hi % REC == SHIFT)   # not record aligned
    first_gen, last_gen = lo // REC, min(TOTAL_READS, (hi + REC - 1) // REC + 1)
    ctx = B.Context(B.ParserConfig(), "generic", 4096, 0, min_record_bytes=256)
    nb = ctx.generate_synthetic_device(TOTAL_READS, 150, 33, 73, "generic", first=first_gen, count=last_gen - first_gen)
    assert nb == (last_gen - first_gen) * REC
    buf = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
    ctx.generate_synthetic_device(TOTAL_READS, 150, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=first_gen, count=last_gen - first_gen)
    off = lo - first_gen * REC
    shard_ptr = buf.data_ptr() + off
    assert shard_ptr % 16 == 0
    # what the neighbours contribute, analytically
    lines_before = 4 * first_gen + (1 if off else 0)                 # the cut sits in a sequence line (line 1 of its record)
    prev_last = int(buf[off - 1].item()) if off else 10
    head = (REC - off) if off else 0
    halo = (REC - SHIFT) if rank + 1 < WORLD else 0                   # the next shard's head: the rest of our last record
    s = ctx.shard_scan(shard_ptr, n)
    assert int(s.n_bytes) == n
    first_owned = first_gen + (1 if off else 0)
    owned_end = (hi + REC - 1) // REC if rank + 1 < WORLD else TOTAL_READS      # records whose header starts before hi
    R = owned_end - first_owned
    assert R == 78_125_000 + (1 if rank == 0 else (-1 if rank == WORLD - 1 else 0))
    # newlines in [lo, hi): closed form from the four newline offsets of a record (15, 166, 168, 319)
    def newlines_before(pos):
        q, r = divmod(pos, REC)
        return 4 * q + sum(1 for o in (15, 166, 168, 319) if o < r)
    assert int(s.n_newlines) == newlines_before(hi) - newlines_before(lo)
    want_first_nl = [p - lo for p in sorted((lo // REC + k) * REC + o for k in (0, 1) for o in (15, 166, 168, 319)) if p >= lo][:4]
    assert [int(x) for x in s.first_nl] == want_first_nl
    ctx.submit_shard(shard_ptr, n, halo, lines_before, prev_last, lo, rank + 1 == WORLD)
    res = ctx.result()
    assert int(res.n_records) == R and res.status == (6 if rank + 1 == WORLD else 0), (int(res.n_records), R, res.status)
    assert int(res.seq_bytes) == int(res.qual_bytes) == 150 * R and int(res.id_bytes) == 14 * R
    assert int(res.bytes_consumed) == head + REC * R
    m = buf[off + head: off + head + REC * R].view(R, REC)
    assert torch.equal(_view(res.d_seq, 150 * R, torch.uint8).view(R, 150), m[:, 16:166])
    assert torch.equal(_view(res.d_qual, 150 * R, torch.uint8).view(R, 150), m[:, 169:319])
    assert torch.equal(_view(res.d_id, 14 * R, torch.uint8).view(R, 14), m[:, 1:15])
    # first and last owned record by name (global record index in the header)
    ids = _view(res.d_id, 14 * R, torch.uint8).view(R, 14)
    assert bytes(ids[0].cpu().numpy()) == b"read_%09d" % first_owned and bytes(ids[-1].cpu().numpy()) == b"read_%09d" % (owned_end - 1)
    r = torch.arange(1, R + 1, dtype=torch.int64, device="cuda")
    res._cumulative()   # (ABI 2: the chunk-cumulative arrays are derived on demand)
    assert torch.equal(_view(res.d_ends, 8 * R, torch.int64), 150 * r)
    assert torch.equal(_view(res.d_id_ends, 8 * R, torch.int64), 14 * r)
    assert torch.equal(_view(res.d_record_end, 8 * R, torch.int64), head + REC * r - 1)
    in_batch = (torch.arange(R, dtype=torch.int64, device="cuda") % 4096) + 1
    assert torch.equal(_view(res.d_batch_ends, 8 * R, torch.int64), 150 * in_batch)
    del r, in_batch, m, ids
    ctx.close()
