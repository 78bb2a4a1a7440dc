"""-m gpu: lifetime and size rules of the boundary.
 * every unaligned bzq_batch_view keeps its own ends storage (a batch held across later next_batch() calls used to read
   the later batch's ends);
 * a chunk's results stay valid until the SECOND following submit (double-buffered output sets), so a consumer kernel on
   another stream overlaps the next chunk's parse;
 * the ingest carries records / batches larger than its reserve (FASTQ: next_batch(n) spanning many chunks; FASTA: a
   record of tens of MB -- the reference has no record-size limit, fasta/parser.mojo:122-172)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _oracle_batches(data: bytes, bs: int):
    sp = O.StreamParser(np.frombuffer(data, dtype=np.uint8), O.make_config(batch_size=bs))
    out = []
    while True:
        b = sp.next_batch(bs)
        if len(b) == 0:
            return out
        out.append(b)


def _same(b, ob):
    assert len(b) == len(ob)
    np.testing.assert_array_equal(b._ends, np.asarray(ob.ends))
    np.testing.assert_array_equal(b._id_ends, np.asarray(ob.id_ends))
    assert b._sequence_bytes.tobytes() == bytes(ob.seq_bytes)
    assert b._quality_bytes.tobytes() == bytes(ob.qual_bytes)
    assert b._id_bytes.tobytes() == bytes(ob.id_bytes)


def test_unaligned_batches_held_before_reading_any_of_them():
    import blazeseq_amd as B
    data = O.generate_synthetic(3000, 20, 180, 0, 40, "sanger").tobytes()   # variable-length reads
    want = _oracle_batches(data, 100)
    p = B.FastqParser(data, batch_size=4096)        # ctx batch size 4096, next_batch(100): never batch aligned after the first
    held = [p.next_batch(100) for _ in range(7)]    # nothing fetched yet
    held += [p.next_batch(37), p.next_batch(100)]
    for b, ob in zip(held[:7], want[:7]):
        _same(b, ob)
    assert len(held[7]) == 37 and len(held[8]) == 100
    d = [b.to_device() for b in held[:3]]           # DeviceFastqBatch.ends of an unaligned view: its own storage as well
    e = np.empty(100, dtype=np.int64)
    for k in (2, 0, 1):
        p._ctx.copy_to_host(e, d[k].ends, 800)
        np.testing.assert_array_equal(e, np.asarray(want[k].ends))


def test_python_surface_batches_of_100_on_variable_length_reads(tmp_path):
    import blazeseq_amd as B
    from blazeseq_amd import pyapi
    data = O.generate_synthetic(2500, 15, 120, 0, 40, "sanger").tobytes()
    path = tmp_path / "v.fastq"
    path.write_bytes(data)
    got = list(pyapi.parser(str(path)).batches)     # list() holds every batch before any of them is read
    want = _oracle_batches(data, 100)
    assert [b.num_records() for b in got] == [len(w) for w in want]
    for b, w in zip(got, want):
        recs = [(r.id, r.sequence, r.quality) for r in b]
        i0 = s0 = 0
        for k, (rid, seq, qual) in enumerate(recs):
            i1, s1 = int(w.id_ends[k]), int(w.ends[k])
            assert (rid.encode("latin-1"), seq.encode("latin-1"), qual.encode("latin-1")) == \
                   (bytes(w.id_bytes[i0:i1]), bytes(w.seq_bytes[s0:s1]), bytes(w.qual_bytes[s0:s1]))
            i0, s0 = i1, s1


def test_results_stay_valid_through_the_next_submit_and_consumers_overlap_it():
    import torch
    import blazeseq_amd as B
    a = O.generate_synthetic(60_000, 150, 150, 0, 40, "sanger")
    b = O.generate_synthetic(50_000, 100, 100, 5, 35, "sanger")
    ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
    da, db = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()
    side = torch.cuda.Stream()
    ctx.set_consumer_stream(side.cuda_stream)
    fa = O.flat_parse(a, O.make_config())
    fb = O.flat_parse(b, O.make_config())

    ctx.submit_device(da.data_ptr(), da.numel(), 0, True)
    ra = ctx.result()
    view_a = ctx.batch_view(100, 50_000)                    # unaligned: its ends live in chunk A's output set
    sums = torch.zeros(50_000, dtype=torch.int64, device="cuda")
    dba = B.DeviceFastqBatch(ctx, view_a)
    dba.quality_sums(sums.data_ptr())                       # on the side stream ...
    ctx.submit_device(db.data_ptr(), db.numel(), 0, True)   # ... while chunk B is parsed on the ctx stream
    rb = ctx.result()
    side.synchronize()
    q = fa.qual_bytes.astype(np.int64) - 33
    cs = np.concatenate([[0], np.cumsum(q)])
    want = cs[fa.ends[100:50_100]] - cs[np.concatenate([[fa.ends[99]], fa.ends[100:50_099]])]
    np.testing.assert_array_equal(sums.cpu().numpy(), want)
    # chunk A's columns are still what they were, chunk B's are B's
    np.testing.assert_array_equal(ra.seq(), fa.seq_bytes)
    np.testing.assert_array_equal(ra.ends(), fa.ends)      # derived on demand (ABI 2) -- here AFTER chunk B was parsed: the set keeps what it takes
    np.testing.assert_array_equal(ra.id_ends(), fa.id_ends)
    np.testing.assert_array_equal(rb.ends(), fb.ends)
    np.testing.assert_array_equal(rb.seq(), fb.seq_bytes)
    np.testing.assert_array_equal(rb.qual(), fb.qual_bytes)
    e = np.empty(50_000, dtype=np.int64)
    ctx.copy_to_host(e, view_a.ends, e.nbytes)
    np.testing.assert_array_equal(e, fa.ends[100:50_100] - fa.ends[99])
    # the third submit recycles chunk A's set
    ctx.submit_device(da.data_ptr(), da.numel(), 0, True)
    rc = ctx.result()
    assert rc.d_seq == ra.d_seq and rc.d_seq != rb.d_seq
    # ... and chunk A (two submits ago) is gone: asking for its cumulative ends now is refused, not answered from chunk C's set
    import copy
    stale = copy.copy(ra); stale.d_ends = None; stale.d_id_ends = None
    with pytest.raises(RuntimeError):
        stale._cumulative()
    # one set only: results are replaced by the next submit
    ctx.set_option("double_buffer", 0)
    ctx.submit_device(db.data_ptr(), db.numel(), 0, True)
    r1 = ctx.result()
    ctx.submit_device(da.data_ptr(), da.numel(), 0, True)
    r2 = ctx.result()
    assert r1.d_seq == r2.d_seq
    np.testing.assert_array_equal(r2.qual(), fa.qual_bytes)
    ctx.close()


def test_fastq_batch_larger_than_the_ingest_reserve(tmp_path):
    """next_batch(n) with n records far beyond one chunk AND beyond the 16 MiB carry reserve: the ingest assembles the
    oversized chunk in a buffer grown to fit (it used to fail with BZQ_ERR_NOMEM)."""
    import blazeseq_amd as B
    data = O.generate_synthetic(90_000, 150, 150, 0, 40, "sanger").tobytes()   # 28 MB
    path = tmp_path / "big.fastq"
    path.write_bytes(data)
    p = B.FastqParser(str(path), batch_size=4096, chunk_bytes=2 << 20)
    b = p.next_batch(80_000)                                                  # 25 MB of records from 2 MiB chunks
    f = O.flat_parse(np.frombuffer(data, dtype=np.uint8), O.make_config())
    assert len(b) == 80_000
    np.testing.assert_array_equal(b._ends, f.ends[:80_000])
    np.testing.assert_array_equal(b._sequence_bytes, f.seq_bytes[:int(f.ends[79_999])])
    rest = p.next_batch(80_000)
    assert len(rest) == 10_000 and not len(p.next_batch(5))
    np.testing.assert_array_equal(rest._quality_bytes, f.qual_bytes[int(f.ends[79_999]):])


def test_fasta_record_larger_than_the_ingest_reserve(tmp_path):
    """A chromosome-sized record (24 MB, 60 columns) through FastaParser(path) with 1 MiB chunks: more than the 16 MiB
    reserve has to be carried from chunk to chunk."""
    import blazeseq_amd as B
    from oracle import fasta as F
    rng = np.random.default_rng(5)
    big = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 24_000_000)
    lines = np.full((big.size // 60, 61), 10, dtype=np.uint8)
    lines[:, :60] = big.reshape(-1, 60)
    data = b">small one\nACGT\nAC\n>chr_big some description\n" + lines.tobytes() + b">after\nGGGG\n"
    path = tmp_path / "big.fasta"
    path.write_bytes(data)
    p = B.FastaParser(str(path), chunk_bytes=1 << 20)
    recs = list(p.records())
    assert [r.id for r in recs] == [b"small one", b"chr_big some description", b"after"]
    assert recs[0].sequence == b"ACGTAC" and recs[2].sequence == b"GGGG"
    assert len(recs[1].sequence) == big.size and recs[1].sequence == big.tobytes()
    p.close()


def test_a_mistyped_path_is_not_fastq_content():
    import blazeseq_amd as B
    with pytest.raises(FileNotFoundError):
        B.FastqParser("/no/such/dir/reads.fastq")
    p = B.FastqParser("@r1\nACGT\n+\nIIII\n")            # a str that IS content
    assert len(p.next_batch(4)) == 1


def test_all_batches_in_one_call_equal_successive_views():
    """bzq_batches(ctx, n) = the batches() iteration over the current chunk: the same structs as successive bzq_batch_view
    calls, for a batch size equal to the ctx's (zero copy, host-cached boundaries) and for one that is not (own storage)."""
    import blazeseq_amd as B
    data = O.generate_synthetic(10_000, 20, 180, 0, 40, "sanger")
    ctx = B.Context(B.ParserConfig(), "generic", 512, 0)
    res = ctx.parse(data, 0, True)
    f = O.flat_parse(data, O.make_config(batch_size=512))
    for bs in (512, 300):
        arr, nb = ctx.batches(bs)
        assert nb == (f.n_records + bs - 1) // bs
        e = np.empty(bs, dtype=np.int64)
        for k in range(nb):
            v = ctx.batch_view(k * bs, bs)
            a = arr[k]
            assert (a.num_records, a.seq_len, a.total_id_bytes, a.qual_buffer, a.sequence_buffer, a.id_buffer, a.first_record) == \
                   (v.num_records, v.seq_len, v.total_id_bytes, v.qual_buffer, v.sequence_buffer, v.id_buffer, v.first_record)
            lo = k * bs
            base = int(f.ends[lo - 1]) if lo else 0
            m = int(a.num_records)
            assert m == min(bs, f.n_records - lo) and int(a.seq_len) == int(f.ends[lo + m - 1]) - base
            ctx.copy_to_host(e[:m], a.ends, 8 * m)
            np.testing.assert_array_equal(e[:m], f.ends[lo:lo + m] - base)
    ctx.close()


def test_batches_of_a_chunk_are_served_while_the_next_chunk_is_parsed():
    """The host's pipeline under the double-buffer contract: result(k) -> submit(k + 1) at once -> walk the batches of chunk k under the
    parse of chunk k + 1 (bench.py's step does exactly this).  bzq_batches / bzq_batch_view / bzq_chunk_cumulative_ends serve the
    chunk the last RESULT described until its output set is written again -- the second submit after its own -- and refuse after."""
    import torch
    import blazeseq_amd as B
    a = O.generate_synthetic(30_000, 20, 180, 0, 40, "sanger")
    b = O.generate_synthetic(21_000, 100, 100, 5, 35, "sanger")
    fa, fb = O.flat_parse(a, O.make_config(batch_size=512)), O.flat_parse(b, O.make_config(batch_size=512))
    ctx = B.Context(B.ParserConfig(), "generic", 512, 0)
    da, db = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()

    def check_batches(f, bs):
        arr, nb = ctx.batches(bs)
        assert nb == (f.n_records + bs - 1) // bs
        e = np.empty(bs, dtype=np.int64)
        for k in range(nb):
            lo, m = k * bs, int(arr[k].num_records)
            base = int(f.ends[lo - 1]) if lo else 0
            assert m == min(bs, f.n_records - lo) and int(arr[k].seq_len) == int(f.ends[lo + m - 1]) - base
            if k in (0, 1, nb // 2, nb - 1):
                ctx.copy_to_host(e[:m], arr[k].ends, 8 * m)
                np.testing.assert_array_equal(e[:m], f.ends[lo:lo + m] - base)
                q = np.empty(int(arr[k].seq_len), dtype=np.uint8)
                ctx.copy_to_host(q, arr[k].qual_buffer, q.size)
                np.testing.assert_array_equal(q, f.qual_bytes[base:base + q.size])

    ctx.submit_device(da.data_ptr(), da.numel(), 0, True)
    ra = ctx.result()
    ctx.submit_device(db.data_ptr(), db.numel(), 0, True)      # chunk B in flight, no result taken
    check_batches(fa, 512)                                     # aligned: zero copy, host-cached boundaries of chunk A's set
    check_batches(fa, 300)                                     # unaligned: own storage in chunk A's set, kernels queue behind B's parse
    np.testing.assert_array_equal(ra.ends(), fa.ends)          # cumulative ends of A derived while B is pending (OutSet::parsed)
    rb = ctx.result()
    check_batches(fb, 512)
    np.testing.assert_array_equal(rb.id_ends(), fb.id_ends)
    # two submits without a result in between: the set the last result (B) lives in is written again -> nothing to serve
    ctx.submit_device(da.data_ptr(), da.numel(), 0, True)
    check_batches(fb, 512)                                     # (one submit later: still B)
    ctx.submit_device(db.data_ptr(), db.numel(), 0, True)
    with pytest.raises(RuntimeError):
        ctx.batches(512)
    rd = ctx.result()
    check_batches(fb, 300)
    # a bzq_chunk from two submits ago names arrays that hold ANOTHER chunk now (same pointers): refused by its serial
    assert rd.raw.chunk_serial == 4 and rb.raw.chunk_serial == 2 and rd.raw.d_batch_ends == rb.raw.d_batch_ends
    import ctypes as C
    from blazeseq_amd import _lib as L
    stale = L.BzqChunk.from_buffer_copy(bytes(rb.raw)); stale.d_ends = None; stale.d_id_ends = None
    assert L.lib().bzq_chunk_cumulative_ends(ctx.h, C.byref(stale)) < 0 and b"no longer alive" in L.lib().bzq_last_error(ctx.h)
    ctx.close()


@pytest.mark.parametrize("kind", ["plain", "bgzf"])
def test_ingest_close_does_not_wait_for_a_callers_other_streams(tmp_path, kind):
    """VERDICT r5 next-7 / ADVICE r4: bzq_ingest_close used to hipDeviceSynchronize() (buffers that go back to the cache skip hipFree's
    implicit wait), i.e. it waited for EVERY stream of the process -- a consumer's long kernel on its own stream included.  It now waits
    for the streams that can still touch the chunk buffers (the parser's; in views mode the consumer's) and the small buffers of the
    BGZF path stay cached instead of going through hipFree / hipHostFree (both wait for the whole device)."""
    import time
    import torch
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    import struct
    import zlib

    def bgzf(data, block=65280):
        out = []
        for i in list(range(0, len(data), block)) + [None]:
            chunk = b"" if i is None else data[i:i + block]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1)
                       + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        return b"".join(out)

    data = O.generate_synthetic(40_000, 150, 150, 0, 40, "sanger").tobytes()
    path = tmp_path / ("r.fastq" if kind == "plain" else "r.fastq.gz")
    path.write_bytes(data if kind == "plain" else bgzf(data))
    ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)

    def run_file():
        ing = B.Ingest(ctx, str(path), chunk_bytes=4 << 20, n_threads=2)
        taken = total = 0
        while True:
            r = ing.next(taken)
            taken = int(r.n_records); total += taken
            if int(r.status) != L.OK:
                break
        assert total == 40_000 and int(r.status) == L.EOF
        return ing

    # (what a close cannot keep in the library's buffer cache goes back to the driver through hipFree, which does wait for the device:
    # empty the cache of whatever earlier tests of this process left in it, and give it room for this file's buffers)
    for key in ("pin_cache_bytes", "dev_cache_bytes"):
        ctx.set_option(key, 0)
        ctx.set_option(key, 2 << 30)
    run_file().close()                       # warm: first-use costs, and the cache holds the buffers
    side = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(100_000_000); torch.cuda.synchronize()
    rate = 100_000_000 / (time.perf_counter() - t0)                  # the spin kernel's counter, ticks per second
    ing = run_file()
    with torch.cuda.stream(side):
        torch.cuda._sleep(int(1.5 * rate))   # a caller's kernel that is nowhere near done: >= ~1 s on its own stream
        busy = torch.cuda.Event(); busy.record(side)
    t0 = time.perf_counter()
    ing.close()
    dt = time.perf_counter() - t0
    still_running = not busy.query()
    side.synchronize()
    total_sleep = time.perf_counter() - t0
    assert still_running and dt < 0.25 * total_sleep, (dt, total_sleep, still_running)   # the close came back while the other stream was still busy
    for key in ("pin_cache_bytes", "dev_cache_bytes"):
        ctx.set_option(key, 0)
        ctx.set_option(key, 1 << 30)
    ctx.close()


def test_views_of_a_chunk_behind_the_next_submit():
    """Views mode hands out offsets INTO the chunk.  Behind the next submit they are served where the chunk's bytes are the caller's
    (bzq_submit_chunk_device: the double-buffer contract covers the offset arrays, the caller keeps its buffer) and refused where they
    sat in the ctx's staging buffer that the next host submit overwrites."""
    import ctypes as C
    import torch
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    a = O.generate_synthetic(20_000, 30, 160, 0, 40, "sanger")
    b = O.generate_synthetic(15_000, 100, 100, 5, 35, "sanger")
    fa = O.flat_parse(a, O.make_config())
    ctx = B.Context(B.ParserConfig(views_only=True), "generic", 4096, 0)
    da, db = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()
    ctx.submit_device(da.data_ptr(), da.numel(), 0, True)
    ra = ctx.result()
    ctx.submit_device(db.data_ptr(), db.numel(), 0, True)          # chunk B in flight
    v = L.BzqDeviceViews()
    assert L.lib().bzq_views(ctx.h, 100, 5000, C.byref(v)) == 0 and v.num_records == 5000 and v.chunk == da.data_ptr()
    e = np.empty(5000, dtype=np.int64)
    ctx.copy_to_host(e, v.record_end, e.nbytes)
    np.testing.assert_array_equal(e, fa.record_end[100:5100])
    ctx.copy_to_host(e, v.seq_start, e.nbytes)
    np.testing.assert_array_equal(e, fa.seq_start[100:5100])
    rb = ctx.result()
    assert int(rb.n_records) == 15_000
    # host submits: the chunk's bytes live in the ctx's staging buffer
    ra2 = ctx.parse(a, 0, True)
    assert L.lib().bzq_views(ctx.h, 0, 10, C.byref(v)) == 0 and v.num_records == 10
    ctx.submit_host(b, 0, True)
    assert L.lib().bzq_views(ctx.h, 0, 10, C.byref(v)) < 0 and b"overwritten" in L.lib().bzq_last_error(ctx.h)
    assert int(ctx.result().n_records) == 15_000
    ctx.close()
