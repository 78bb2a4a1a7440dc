"""-m gpu: the native ingest pipeline (bzq_ingest_*: reader threads -> pinned double buffers -> device -> chunk parser)
against the oracle, through the C ABI.  Replaces FileReader + BufferedReader refills (io/readers.mojo:86-137,
io/buffered.mojo:239-290) for plain files, so the parity bar is: same records, same batches, same terminal event as
the reference-algorithm streaming parser reading the same file."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from fastq_fuzz import rand_stream

pytestmark = pytest.mark.gpu


def _records_of(ctx, res, first, count):
    """(id, seq, qual) of records [first, first+count) of a parsed chunk, via bzq_batch_view + copy to host."""
    import blazeseq_amd as B
    if count == 0:
        return []
    fb = B.FastqBatch(ctx, ctx.batch_view(first, count))
    return [(r.id, r.sequence, r.quality) for r in fb.to_records()]


def _oracle_records(f):
    out, e0, i0 = [], 0, 0
    for r in range(f.n_records):
        e1, i1 = int(f.ends[r]), int(f.id_ends[r])
        out.append((f.id_bytes[i0:i1].tobytes(), f.seq_bytes[e0:e1].tobytes(), f.qual_bytes[e0:e1].tobytes()))
        e0, i0 = e1, i1
    return out


@pytest.mark.parametrize("seed", range(6))
def test_ingest_chunks_with_partial_takes_match_flat_oracle(seed, tmp_path):
    """Tiny chunks (4-20 KiB), the caller takes all / some / none of each chunk's records: the concatenation of what
    was taken equals the one-shot parse of the file, and the stream ends with the same event."""
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    rng = np.random.default_rng(1000 + seed)
    dirty = 0.0 if seed < 4 else 0.002
    data = rand_stream(rng, n_records=int(rng.integers(400, 1500)), max_len=120, dirty=dirty, tail=[0, 1, 2, 3, 0, 5][seed])
    path = tmp_path / f"s{seed}.fastq"
    path.write_bytes(data)
    ocfg = O.make_config(batch_size=4096)
    f = O.flat_parse(np.frombuffer(data, dtype=np.uint8), ocfg, is_eof=True)
    want = _oracle_records(f)
    ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
    ing = B.Ingest(ctx, str(path), chunk_bytes=[4096, 8192, 20480][seed % 3], n_threads=3)
    got, taken, status = [], 0, L.OK
    for it in range(100000):
        res = ing.next(taken)
        n = int(res.n_records)
        status = int(res.status)
        if status != L.OK:
            take = n                      # terminal chunk: everything it delivers
        else:
            mode = int(rng.integers(0, 4))
            take = n if mode < 2 else (int(rng.integers(0, n + 1)) if mode == 2 else 0)
        got += _records_of(ctx, res, 0, take)
        taken = take
        if status != L.OK:
            break
    assert got == want, (len(got), len(want))
    assert status == f.term_code, (status, f.term_code)
    st = ing.stats()
    assert st.file_bytes == len(data) and st.records == len(want)
    # a stream that ends in a failing record stops the readers early; otherwise every byte was fetched once
    assert st.bytes_read == len(data) if f.consumed + 4096 >= len(data) else st.bytes_read <= len(data)
    # after the terminal chunk the ingest keeps answering with that code and no records
    again = ing.next(0)
    assert int(again.status) == status and int(again.n_records) == 0
    ing.close()


def test_parser_on_a_path_uses_the_ingest_and_matches_the_streaming_oracle(tmp_path):
    import blazeseq_amd as B
    data = O.generate_synthetic(30_000, 50, 150, 0, 40, "sanger")
    path = tmp_path / "syn.fastq"
    path.write_bytes(bytes(data))
    ref = [b for b in O.StreamParser(data, O.make_config(batch_size=1000)).batches()]
    for chunk in (1 << 16, 300_000, 1 << 28):
        p = B.FastqParser(str(path), batch_size=1000, chunk_bytes=chunk)
        assert p._ingest is not None
        got = list(p.batches())
        assert [len(b) for b in got] == [len(b) for b in ref]
        for g, r in zip(got, ref):
            assert g._ends.tolist() == r.ends and g._id_ends.tolist() == r.id_ends
            assert g._sequence_bytes.tobytes() == r.seq_bytes and g._quality_bytes.tobytes() == r.qual_bytes
            assert g._id_bytes.tobytes() == r.id_bytes
    # the Reader-style loop (file object) and the ingest give the same thing
    with open(path, "rb") as fh:
        got2 = list(B.FastqParser(fh, batch_size=1000, chunk_bytes=1 << 16).batches())
    assert [b._sequence_bytes.tobytes() for b in got2] == [r.seq_bytes for r in ref]


def test_ingest_error_in_a_late_chunk_reports_the_global_record_number(tmp_path):
    import blazeseq_amd as B
    recs = [b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(5000)]
    recs[3777] = b"@r3777\nACGTACGTAC\n+\nIIIIIIIII\n"   # quality one byte short
    data = b"".join(recs)
    path = tmp_path / "bad.fastq"
    path.write_bytes(data)
    sp = O.StreamParser(np.frombuffer(data, dtype=np.uint8), O.make_config(batch_size=100))
    n_ok = 0
    with pytest.raises(Exception) as ref_err:
        while True:
            b = sp.next_batch(100)
            if len(b) == 0:
                break
            n_ok += len(b)
    p = B.FastqParser(str(path), batch_size=100, chunk_bytes=1 << 16)
    n = 0
    with pytest.raises(B.ParseError) as err:
        while True:
            b = p.next_batch(100)
            if len(b) == 0:
                break
            n += len(b)
    assert n == n_ok == 3700
    assert b"Record number: 3778" in err.value.message
    assert err.value.message.decode("latin-1") == str(ref_err.value)


def test_ingest_empty_and_missing_files(tmp_path):
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    empty = tmp_path / "empty.fastq"
    empty.write_bytes(b"")
    ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
    ing = B.Ingest(ctx, str(empty))
    res = ing.next(0)
    assert int(res.status) == L.EOF and int(res.n_records) == 0
    ing.close()
    assert list(B.FastqParser(str(empty)).batches()) == []
    with pytest.raises(RuntimeError, match="cannot open"):
        B.Ingest(ctx, str(tmp_path / "nope.fastq"))


# ---- compressed input (the reference's GZFile / RapidgzipReader, io/readers.mojo:283-443) -------------------------

def _bgzf(data: bytes, block: int = 65280) -> bytes:
    """BGZF writer (SAM spec 4.1): independent gzip members with a 'BC' extra subfield + the empty EOF block."""
    import struct
    import zlib
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 18 + len(body) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                   + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


@pytest.mark.parametrize("kind", ["gzip", "gzip_multi_member", "bgzf"])
def test_compressed_files_parse_like_the_plain_file(kind, tmp_path):
    import gzip
    import blazeseq_amd as B
    data = bytes(O.generate_synthetic(20_000, 50, 150, 0, 40, "sanger"))
    if kind == "gzip":
        comp = gzip.compress(data, 1)
    elif kind == "gzip_multi_member":
        cut = len(data) // 3 + 11   # members split in the middle of a record
        comp = gzip.compress(data[:cut], 1) + gzip.compress(data[cut:2 * cut], 6) + gzip.compress(data[2 * cut:], 1)
    else:
        comp = _bgzf(data)
    path = tmp_path / ("reads.fastq.gz")
    path.write_bytes(comp)
    ref = [b for b in O.StreamParser(np.frombuffer(data, dtype=np.uint8), O.make_config(batch_size=1000)).batches()]
    for chunk, gpu_inflate in ((1 << 16, True), (1 << 20, True), (1 << 28, True), (1 << 20, False)):
        # (gpu_inflate: the device decoders of bzq_inflate.hpp / bzq_gzip.hpp, the default; without it BGZF blocks inflate on
        # the reader threads and any other gzip file through zlib's gzread, like the reference's GZFile)
        p = B.FastqParser(str(path), batch_size=1000, chunk_bytes=chunk, reader_threads=3, gpu_inflate=gpu_inflate)
        got = list(p.batches())
        assert [len(b) for b in got] == [len(b) for b in ref]
        for g, r in zip(got, ref):
            assert g._sequence_bytes.tobytes() == r.seq_bytes and g._quality_bytes.tobytes() == r.qual_bytes
            assert g._id_bytes.tobytes() == r.id_bytes and g._ends.tolist() == r.ends
    # the reference's Python surface accepts .gz paths (python/blazeseq/__init__.py:267-290)
    assert sum(1 for _ in B.parser(str(path)).records) == 20_000


@pytest.mark.parametrize("fifo_kib,chunk", [(192, 1 << 16), (1024, 1 << 16), (3072, 1 << 20)])
def test_a_gz_stream_through_the_wrap_around_of_its_fifo(fifo_kib, chunk, tmp_path, monkeypatch):
    """The decoder's output waits in ONE device buffer: chunks leave at its head, pieces arrive behind what waits, and what waits
    moves to the front when less than half the buffer is free (bzq_ingest.hpp gz_fill_fifo; round 5: the second buffer of the same
    size is gone).  With the default size (6 chunks, at least 64 MiB) a test file never gets there: BZQ_GZ_FIFO_KIB (read by
    bzq_ingest_open) makes it 3 chunks to 3 MiB, so that 6.6 MB of FASTQ walk through dozens of moves -- and through pieces whose
    output does not fit what is free (the decoder keeps the rest and is asked again).  Every batch against the oracle."""
    import gzip
    import blazeseq_amd as B
    data = bytes(O.generate_synthetic(20_000, 50, 150, 0, 40, "sanger"))
    cut = len(data) // 5 + 7
    comp = b"".join(gzip.compress(data[i:i + cut], 6 if (i // cut) % 2 else 1) for i in range(0, len(data), cut))
    path = tmp_path / "reads.fastq.gz"
    path.write_bytes(comp)
    ref = [b for b in O.StreamParser(np.frombuffer(data, dtype=np.uint8), O.make_config(batch_size=1000)).batches()]
    monkeypatch.setenv("BZQ_GZ_FIFO_KIB", str(fifo_kib))
    p = B.FastqParser(str(path), batch_size=1000, chunk_bytes=chunk, reader_threads=3)
    got = list(p.batches())
    assert [len(b) for b in got] == [len(b) for b in ref]
    for g, r in zip(got, ref):
        assert g._sequence_bytes.tobytes() == r.seq_bytes and g._quality_bytes.tobytes() == r.qual_bytes
        assert g._id_bytes.tobytes() == r.id_bytes and g._ends.tolist() == r.ends


def test_truncated_gzip_is_a_runtime_error_not_a_parse_result(tmp_path):
    import gzip
    import blazeseq_amd as B
    data = bytes(O.generate_synthetic(5_000, 100, 100, 0, 40, "sanger"))
    for comp in (gzip.compress(data, 1), _bgzf(data)):
        path = tmp_path / "cut.fastq.gz"
        path.write_bytes(comp[: len(comp) // 2])
        for gpu_inflate in (True, False):
            with pytest.raises(RuntimeError, match="gzread|BGZF|gzip"):
                list(B.FastqParser(str(path), batch_size=1000, gpu_inflate=gpu_inflate).batches())
    # a damaged payload inside a whole block: the device decoder refuses it (bad code / distance / size), like zlib does
    comp = bytearray(_bgzf(data))
    for off in range(400, 4000, 97):
        comp[off] ^= 0x5A
    path = tmp_path / "bad.fastq.gz"
    path.write_bytes(bytes(comp))
    for gpu_inflate in (True, False):
        with pytest.raises(RuntimeError, match="BGZF"):
            list(B.FastqParser(str(path), batch_size=1000, gpu_inflate=gpu_inflate).batches())


@pytest.mark.parametrize("defer", ["0", "1"])
def test_a_damaged_member_in_an_early_piece_fails_the_file(defer, tmp_path, monkeypatch):
    """Round 5: through the ingest a piece's member checks (CRC-32, ISIZE) are made at the start of the NEXT bzq_gzip_decode call (option
    defer_verify: the call that launched the next piece's decoders returns without waiting for its own last kernels).  A member whose
    trailer is wrong in the middle of a file of many pieces still fails the file -- like gzread, bytes of the damaged member may have been
    delivered before the error is -- and an undamaged file gives the same records either way (BZQ_GZ_DEFER overrides the option)."""
    import gzip
    import blazeseq_amd as B
    data = bytes(O.generate_synthetic(30_000, 100, 100, 0, 40, "sanger"))
    cut = len(data) // 12 + 5
    members = [gzip.compress(data[i:i + cut], 6) for i in range(0, len(data), cut)]
    monkeypatch.setenv("BZQ_GZ_DEFER", defer)
    path = tmp_path / "good.fastq.gz"
    path.write_bytes(b"".join(members))
    assert sum(len(b) for b in B.FastqParser(str(path), batch_size=1000, chunk_bytes=1 << 18, reader_threads=2).batches()) == 30_000
    for which, field in ((2, -8), (5, -4)):   # a wrong CRC-32 in member 2, a wrong ISIZE in member 5
        bad = [bytearray(m) for m in members]
        bad[which][field] ^= 0x01
        path = tmp_path / "bad.fastq.gz"
        path.write_bytes(b"".join(bytes(m) for m in bad))
        with pytest.raises(RuntimeError, match="CRC-32|length"):
            list(B.FastqParser(str(path), batch_size=1000, chunk_bytes=1 << 18, reader_threads=2).batches())


# ---- the reference's window at the end of a stream that came in several chunks ----------------------------------------

@pytest.mark.parametrize("source", ["memory", "file"])
def test_tail_of_a_multi_chunk_stream_is_judged_with_the_reference_window(source, tmp_path):
    """An unterminated last record is accepted when the reference's window already reaches EOF and refused with
    BUFFER_EXCEEDED when the record straddles the window's end (io/buffered.mojo:239-290, parser.mojo:451-522); junk
    flips between UNEXPECTED_EOF and BUFFER_EXCEEDED the same way.  Where that window sits depends on every record since
    the first byte, so a stream parsed in several chunks keeps a log of its record ends (option "records_before").  A
    small buffer_capacity makes the refusing alignment common enough to meet it in a few dozen streams."""
    import blazeseq_amd as B
    cap = 4096
    classes = set()
    for seed in range(48):
        rng = np.random.default_rng(8800 + seed)
        tail = [1, 2, 3, 5][seed % 4]
        data = rand_stream(rng, n_records=int(rng.integers(900, 1500)), max_len=150, dirty=0.0, tail=tail)
        ocfg = O.make_config(buffer_capacity=cap, batch_size=256)
        sp = O.StreamParser(np.frombuffer(data, dtype=np.uint8), ocfg)
        want, werr = [], None
        while True:
            try:
                b = sp.next_batch(256)
            except O.OracleError as e:
                werr = (e.code, str(e))
                break
            if len(b) == 0:
                break
            want.append((b.seq_bytes, b.ends))
        if source == "file":
            path = tmp_path / "s.fastq"
            path.write_bytes(data)
            src = str(path)
        else:
            src = data
        p = B.FastqParser(src, batch_size=256, config=B.ParserConfig(buffer_capacity=cap), chunk_bytes=1 << 16)
        got, gerr = [], None
        while True:
            try:
                b = p.next_batch(256)
            except B.ParseError as e:
                gerr = (e.code, e.message.decode("latin-1"))
                break
            if len(b) == 0:
                break
            got.append((b._sequence_bytes.tobytes(), b._ends.tolist()))
        assert len(data) > (1 << 16) + 4096                    # really several chunks
        assert got == want, (seed, len(got), len(want))
        assert gerr == werr, (seed, gerr, werr)
        classes.add(werr[0] if werr else 0)
    assert len(classes) >= 3, classes                          # accepted, refused (BUFFER_EXCEEDED) and UNEXPECTED_EOF all met


@pytest.mark.parametrize("chunk", [1 << 16, 1 << 20])
def test_o_direct_reads_give_the_same_stream(chunk, tmp_path):
    """Option "ingest_direct": whole 4 KiB blocks read O_DIRECT into the pinned buffers, the tail and
    filesystems without O_DIRECT buffered.  Same records either way; the stats say which path the file took."""
    import blazeseq_amd as B
    data = O.generate_synthetic(40_000, 30, 170, 0, 40, "sanger")   # size not a multiple of 4096
    assert data.size % 4096
    path = tmp_path / "direct.fastq"
    path.write_bytes(bytes(data))
    f = O.flat_parse(data, O.make_config())
    outs = []
    for direct in (0, 1):
        ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
        ctx.set_option("ingest_direct", direct)
        ing = B.Ingest(ctx, str(path), chunk, 3)
        seqs, n, taken = [], 0, 0
        while True:
            res = ing.next(taken)
            taken = int(res.n_records)
            n += taken
            seqs.append(res.seq().tobytes())
            if res.status != 0:
                assert res.status == 6
                break
        st = ing.stats()
        assert st.direct_io in (0, 1) and (direct or st.direct_io == 0) and int(st.bytes_read) == data.size
        outs.append((n, b"".join(seqs), int(st.direct_io), int(st.numa_node)))
        ing.close(); ctx.close()
    assert outs[0][0] == outs[1][0] == f.n_records
    assert outs[0][1] == outs[1][1] == f.seq_bytes.tobytes()
    print("direct_io:", outs[1][2], "numa node:", outs[1][3])


@pytest.mark.parametrize("kind", ["plain", "gzip", "bgzf"])
def test_buffers_of_a_closed_file_serve_the_next_open(kind, tmp_path):
    """Round 4 (bzq_bufcache.hpp): the pinned / device chunk buffers of a closed stream stay in a process-wide cache; the next open
    takes them from there -- with whatever bytes the last file left in them -- and must deliver its own file's records; options
    pin_cache_bytes / dev_cache_bytes = 0 give everything back to the driver."""
    import gzip
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    files = []
    for i, (n, lo, hi) in enumerate([(30_000, 50, 150), (9_000, 20, 400), (30_000, 50, 150)]):
        data = bytes(O.generate_synthetic(n, lo, hi, 0, 40, "sanger"))
        if i == 2:
            data = data.replace(b"A", b"C")   # the same sizes as file 0, different bytes
        comp = data if kind == "plain" else gzip.compress(data, 1) if kind == "gzip" else _bgzf(data)
        path = tmp_path / f"reads{i}.fastq{'' if kind == 'plain' else '.gz'}"
        path.write_bytes(comp)
        files.append((str(path), data))
    probe = B.Context(B.ParserConfig(), "generic", 1000, 0)
    q = lambda key: L.lib().bzq_set_option(probe.h, key.encode(), 0)
    for key in ("pin_cache_bytes", "dev_cache_bytes"):
        probe.set_option(key, 0)
    assert q("buf_cache_held_mb") == 0
    probe.set_option("pin_cache_bytes", 2 << 30)
    probe.set_option("dev_cache_bytes", 8 << 30)
    hits0 = q("buf_cache_hits")
    for rnd in range(2):
        for path, data in files:
            ref = [b for b in O.StreamParser(np.frombuffer(data, dtype=np.uint8), O.make_config(batch_size=1000)).batches()]
            p = B.FastqParser(path, batch_size=1000, chunk_bytes=4 << 20, reader_threads=3)
            got = list(p.batches())
            assert [len(b) for b in got] == [len(b) for b in ref]
            for g, r in zip(got, ref):
                assert g._sequence_bytes.tobytes() == r.seq_bytes and g._quality_bytes.tobytes() == r.qual_bytes
                assert g._id_bytes.tobytes() == r.id_bytes and g._ends.tolist() == r.ends
            p.close() if hasattr(p, "close") else None
            del p
    import gc
    gc.collect()
    assert q("buf_cache_hits") >= hits0 + 6, (hits0, q("buf_cache_hits"))   # at least the three slots' pinned + device buffers, once
    assert q("buf_cache_held_mb") > 0
    probe.set_option("pin_cache_bytes", 0)
    probe.set_option("dev_cache_bytes", 0)
    assert q("buf_cache_held_mb") == 0
    probe.set_option("pin_cache_bytes", 1 << 30)   # (the defaults again)
    probe.set_option("dev_cache_bytes", 1 << 30)
    probe.close()
