"""-m gpu: device-side consumers of a DeviceFastqBatch (bzq_consumers.hpp) against restatements of the reference's
example code: the nw_gpu kernel (examples/nw_gpu/kernels.mojo:21-89), FastqRecord.phred_scores (record.mojo:340-346),
and byte histograms of the columns.  Integer work: results must be identical."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def nw_kernel_restated(ref: bytes, query: bytes) -> int:
    """examples/nw_gpu/kernels.mojo:21-89, line for line in Python: two DP rows, match +1, mismatch -1, gap -1,
    0 when either length exceeds 256."""
    ref_len, query_len = len(ref), len(query)
    if query_len > 256 or ref_len > 256:
        return 0
    prev = [-i for i in range(ref_len + 1)]
    for j in range(1, query_len + 1):
        curr = [-j] + [0] * ref_len
        for i in range(1, ref_len + 1):
            diag = prev[i - 1] + (1 if ref[i - 1] == query[j - 1] else -1)
            best = max(diag, prev[i] - 1, curr[i - 1] - 1)
            curr[i] = best
        prev = curr
    return prev[ref_len]


def _batch_from(data: bytes, n):
    import blazeseq_amd as B
    p = B.FastqParser(data, batch_size=n)
    b = p.next_batch(n)
    return p, b, b.to_device()


@pytest.mark.parametrize("ref_len", [0, 1, 17, 40, 64, 65, 130, 256, 300])
def test_nw_scores_match_the_example_kernel(ref_len):
    import torch
    rng = np.random.default_rng(ref_len)
    ref = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), ref_len))
    recs = []
    for i in range(70):
        L = int(rng.integers(0, 60)) if i % 7 else [0, 1, 64, 255, 256, 257, 300][i // 7 % 7]
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L))
        if i % 5 == 0 and L and ref_len:   # reads related to the reference score high
            s = (ref * (L // ref_len + 1))[:L]
        recs.append(b"@r%d\n" % i + s + b"\n+\n" + b"I" * L + b"\n")
    p, b, d = _batch_from(b"".join(recs), len(recs))
    scores = torch.full((d.num_records,), -12345, dtype=torch.int32, device="cuda")
    d.nw_scores(ref, scores.data_ptr())
    torch.cuda.synchronize()
    want = [nw_kernel_restated(ref, r.sequence) for r in b.to_records()]
    assert scores.cpu().tolist() == want


def test_quality_sums_and_histograms_match_numpy():
    import torch
    import blazeseq_amd as B
    data = O.generate_synthetic(20_000, 30, 200, 0, 40, "sanger")
    p = B.FastqParser(data, batch_size=4096)
    seen = 0
    for b in p.batches():
        d = b.to_device()
        sums = torch.empty(d.num_records, dtype=torch.int64, device="cuda")
        d.quality_sums(sums.data_ptr())
        hs, hq = d.histogram("sequence"), d.histogram("quality")
        torch.cuda.synchronize()
        recs = b.to_records()
        want = [int(np.frombuffer(r.quality, dtype=np.uint8).astype(np.int64).sum()) - 33 * len(r.quality) for r in recs]
        assert sums.cpu().tolist() == want
        np.testing.assert_array_equal(hs, np.bincount(b._sequence_bytes, minlength=256).astype(np.uint64))
        np.testing.assert_array_equal(hq, np.bincount(b._quality_bytes, minlength=256).astype(np.uint64))
        seen += len(recs)
    assert seen == 20_000


def test_consumers_on_an_empty_batch_and_odd_sizes():
    import torch
    p, b, d = _batch_from(b"@a\nACGTA\n+\n!!!!!\n", 4)
    s = torch.zeros(1, dtype=torch.int32, device="cuda")
    d.nw_scores(b"ACGTA", s.data_ptr())
    assert s.item() == 5
    assert int(d.histogram("sequence")[ord("A")]) == 2 and int(d.histogram("quality")[ord("!")]) == 5
    q = torch.zeros(1, dtype=torch.int64, device="cuda")
    d.quality_sums(q.data_ptr()); torch.cuda.synchronize()
    assert q.item() == 0


def test_gc_counts_per_record_on_fastq_and_fasta_columns():
    """bzq_column_gc_counts: G/C bases per record of any device byte column delimited by inclusive running sums -- a
    FastqBatch's sequence column (short reads: the block kernel; long reads: the wave kernel) and a FASTA chunk's."""
    import ctypes as C
    import torch
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    from oracle import fasta as F

    def gc_of(seq_bytes, ends):
        isgc = np.isin(seq_bytes, np.frombuffer(b"GCgc", dtype=np.uint8)).astype(np.int64)
        cs = np.concatenate([[0], np.cumsum(isgc)])
        e = np.concatenate([[0], ends])
        return cs[e[1:]] - cs[e[:-1]]

    ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
    for lo, hi, n in ((1, 300, 20_000), (2000, 9000, 600)):
        data = O.generate_synthetic(n, lo, hi, 0, 40, "sanger")
        f = O.flat_parse(data, O.make_config())
        t = torch.from_numpy(data.copy()).cuda()
        ctx.submit_device(t.data_ptr(), t.numel(), 0, True)
        res = ctx.result()
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        res._cumulative()   # the whole chunk as one column: its chunk-cumulative ends (derived on demand since ABI 2)
        assert L.lib().bzq_column_gc_counts(ctx.h, C.c_void_p(res.d_seq), C.c_void_p(res.d_ends), n, int(res.seq_bytes), C.c_void_p(out.data_ptr())) == 0
        assert np.array_equal(out.cpu().numpy(), gc_of(f.seq_bytes, f.ends))
    # a FASTA chunk: mixed case, N and gaps in the sequences, an empty-ish tail record
    fa = B.FastaContext()
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"ACGTacgtN-", dtype=np.uint8)
    data = b"".join(b">r%d\n" % i + alphabet[rng.integers(0, 10, size=int(rng.integers(1, 500)))].tobytes() + b"\n" for i in range(5000))
    r = fa.parse(data, len(data), True)
    w = F.flat_parse(data)
    out = torch.empty(int(r.n_records), dtype=torch.int64, device="cuda")
    assert L.lib().bzq_column_gc_counts(ctx.h, C.c_void_p(r.d_seq_bytes), C.c_void_p(r.d_seq_ends), int(r.n_records), int(r.seq_bytes), C.c_void_p(out.data_ptr())) == 0
    assert np.array_equal(out.cpu().numpy(), gc_of(w.seq_bytes, w.seq_ends))
    fa.close()
    ctx.close()


def test_quality_distribution_per_read_position():
    """counts[p, v] against numpy on variable-length reads (lengths 1..240 and a few empty), positions beyond max_positions
    ignored, every record's byte at every position counted exactly once."""
    import blazeseq_amd as B
    rng = np.random.default_rng(11)
    recs, quals = [], []
    for i in range(9000):
        L = int(rng.integers(0, 241)) if i % 50 else 0
        q = bytes(rng.integers(33, 127, L).astype(np.uint8))
        quals.append(q)
        recs.append(b"@r%d\n" % i + b"A" * L + b"\n+\n" + q + b"\n")
    p = B.FastqParser(b"".join(recs), batch_size=len(recs))
    b = p.next_batch(len(recs))
    d = b.to_device()
    for max_pos in (1, 64, 100, 240, 300):
        got = d.quality_by_position(max_pos)
        want = np.zeros((max_pos, 128), dtype=np.uint64)
        for q in quals:
            a = np.frombuffer(q, dtype=np.uint8)[:max_pos]
            np.add.at(want, (np.arange(a.size), a), 1)
        np.testing.assert_array_equal(got, want)
    assert int(d.quality_by_position(240).sum()) == sum(len(q) for q in quals)


@pytest.mark.parametrize("guard", [0, 1])
def test_pipeline_consumers_stay_on_the_device_and_equal_the_cpu_twin(tmp_path, guard):
    """The reference's GPU use case as a pipeline (examples/nw_gpu/execution.mojo:100-130): file -> bzq_ingest_next -> batches ->
    bzq_batch_nw_scores_dev + bzq_batch_quality_by_position_acc on the consumer stream, nothing synchronised per batch, chunk k's
    consumers running under the parse of chunk k + 1.  The two-chunk lifetime rule is kept by a host-side event per chunk (guard 0) or
    by the library on the device (option consumer_guard = 1: no event, no host wait).  Scores and the accumulated per-position table
    must equal the oracle's CPU twin (orc_pipeline_run) over the same file."""
    import ctypes as C
    import torch
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    REF = b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"
    data = O.generate_synthetic(60_000, 20, 200, 0, 40, "sanger")
    path = tmp_path / "p.fastq"
    path.write_bytes(data.tobytes())
    n_want, counts_want, ss_want = O.pipeline_run(data, O.make_config(buffer_capacity=64 * 1024, batch_size=4096), REF, 150)
    f = O.flat_parse(data, O.make_config())

    ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
    side = torch.cuda.Stream()
    ctx.set_consumer_stream(side.cuda_stream)
    ctx.set_option("consumer_guard", guard)
    d_ref = torch.frombuffer(bytearray(REF), dtype=torch.uint8).cuda()
    d_counts = torch.zeros(150 * 128, dtype=torch.int64, device="cuda")
    d_scores = torch.full((n_want,), -99999, dtype=torch.int32, device="cuda")
    ing = B.Ingest(ctx, str(path), chunk_bytes=1 << 20, n_threads=2)      # ~12 chunks
    taken = total = 0
    events = []
    while True:
        r = ing.next(taken)
        taken = int(r.n_records)
        if not guard and len(events) >= 1:
            events[-1].synchronize()    # the consumers of the chunk before this one are through before the NEXT next() submits again
        arr, nb = ctx.batches(4096)
        for k in range(nb):
            assert L.lib().bzq_batch_nw_scores_dev(ctx.h, C.byref(arr[k]), C.c_void_p(d_ref.data_ptr()), len(REF),
                                                   C.c_void_p(d_scores.data_ptr() + 4 * (total + k * 4096))) == 0
            assert L.lib().bzq_batch_quality_by_position_acc(ctx.h, C.byref(arr[k]), 150, C.c_void_p(d_counts.data_ptr())) == 0
        ev = torch.cuda.Event(); ev.record(side); events.append(ev)
        total += taken
        if int(r.status) != L.OK:
            break
    ing.close()
    side.synchronize()
    assert total == n_want == 60_000 and len(events) > 5
    np.testing.assert_array_equal(d_counts.cpu().numpy().reshape(150, 128).astype(np.uint64), counts_want)
    scores = d_scores.cpu().numpy()
    assert int(scores.astype(np.int64).sum()) == ss_want
    e = np.concatenate([[0], f.ends])
    for r_ in (0, 1, 4095, 4096, 31_234, 59_999):
        assert int(scores[r_]) == O.nw_score(REF, f.seq_bytes[e[r_]:e[r_ + 1]].tobytes())
    # refused, not clamped: a reference longer than the kernel's limit
    assert L.lib().bzq_batch_nw_scores_dev(ctx.h, C.byref(arr[0]), C.c_void_p(d_ref.data_ptr()), 300, C.c_void_p(d_scores.data_ptr())) < 0
    ctx.close()
