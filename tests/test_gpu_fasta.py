"""-m gpu: the FASTA path (bzq_fasta_* over csrc/bzq_fasta.hpp) against the oracle (oracle/fasta.py), through the C ABI.
Bit-exact: columns, ends, '>' offsets, the carry point of a chunk, the first error and its text."""
import os

import numpy as np
import pytest

from oracle import fasta as F
from fasta_fuzz import rand_fasta, rand_soup

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fasta")
TILE = 16384


@pytest.fixture(scope="module")
def ctxs():
    import blazeseq_amd as B
    made = {}

    def get(check_ascii=False, cap=256 * 1024):
        key = (check_ascii, cap)
        if key not in made:
            made[key] = B.FastaContext(B.FastaParserConfig(check_ascii, cap))
        return made[key]
    yield get
    for c in made.values():
        c.close()


def check_chunk(ctx, data: bytes, is_eof=True, bases=(0, 0, 0)):
    """One bzq_fasta_parse against the flat oracle with the same arguments."""
    from blazeseq_amd import _lib as L
    pos, lines, recs = bases
    want = F.flat_parse(data, ctx.config.check_ascii, ctx.config.line_capacity, is_eof, recs, lines, pos)
    res = ctx.parse(data, len(data), is_eof, pos, lines, recs)
    status = int(res.status)
    assert status == want.status, (status, want.status, want.message, ctx.error_text())
    assert int(res.n_records) == want.n_records
    idb, ide, sqb, sqe, hp = ctx.columns(res)
    assert ide.tolist() == want.id_ends.tolist()
    assert sqe.tolist() == want.seq_ends.tolist()
    assert idb.tobytes() == want.id_bytes.tobytes()
    assert sqb.tobytes() == want.seq_bytes.tobytes()
    assert hp.tolist() == want.hdr_pos.tolist()
    if status in (L.OK, L.FASTA_NEED_MORE):
        assert int(res.bytes_consumed) == want.consumed and int(res.lines_consumed) == want.lines_consumed
    elif status != L.EOF:
        assert ctx.error_text().decode("latin-1") == want.message
        assert (int(res.err_record_number), int(res.err_line_number), int(res.err_file_position)) == \
            (want.err_record_number, want.err_line_number, want.err_file_position)
    return want


KATS = [b">id1\nACGT\n", b">id1\nAC\nGT\n", b">id1\nACGT\n>id2\nTTAA\n", b">id1\nACGT", b"ACGT\n>id1\nACGT\n", b"ACGTACGT\n",
        b">id\x80\nACGT\n", b">id1\nAC\x80GT\n", b">id1\nAC\nGT\n>id2\nTT\nAA\n", b"\n\n\n>id1\nACGT\n", b">id1\nACGT\n\n\n>id2\nTTAA\n",
        b">id1\r\nACGT\r\n", b">id1\r\nACGT\r\n>id2\r\nTTAA\r\n", b">  spaced_id\nACGT\n", b">seq_id   \nACGT\n", b">\ttab_id\t\nACGT\n",
        b">\nACGT\n", b">id1\nA\n", b">id1\nA\nC\nG\nT\nA\nC\nG\nT\n", b">id1\nACG\nTTA", b">id1\n", b">id1\n>id2\nACGT\n",
        b">id1\nACGT\n>id2\n>id3\nGGGG\n", b"", b"\n\n   \n\t\n", b">a>b\nAC>GT\n  >c  d \n A C \n", b" ", b">", b"\n", b">\n>\n", b"x"]


@pytest.mark.parametrize("check", [False, True])
def test_reference_kats_through_the_c_abi(ctxs, check):   # tests/fasta/test_fasta_parser.mojo, see test_oracle_fasta_kats.py
    ctx = ctxs(check)
    for data in KATS:
        check_chunk(ctx, data, True)
        check_chunk(ctx, data, False)
        check_chunk(ctx, data, True, bases=(1000, 50, 7))


def test_biopython_files(ctxs):   # tests/fasta/test_fasta_parser_correctness.mojo
    ctx = ctxs(True)
    for name in sorted(os.listdir(GOLD)):
        if name.endswith(".md"):
            continue
        with open(os.path.join(GOLD, name), "rb") as fh:
            w = check_chunk(ctx, fh.read(), True)
        # the two files with leading comment lines are not FASTA to the reference either (it skips them in its tests)
        assert (w.status == F.NO_HEADER) if name in ("aster_blast.pro", "aster_pearson.pro") else (w.status == F.EOF and w.n_records >= 1)


@pytest.mark.parametrize("seed", range(12))
def test_random_fasta_streams(ctxs, seed):
    rng = np.random.default_rng(100 + seed)
    ctx = ctxs(bool(seed & 1))
    for _ in range(12):
        data = rand_fasta(rng, int(rng.integers(1, 400)), int(rng.choice([20, 70, 300])), int(rng.integers(1, 8)),
                          dirty=float(rng.choice([0, 0.01, 0.1, 0.4])), crlf=bool(rng.integers(0, 2)), tail_newline=bool(rng.integers(0, 2)),
                          lead_blank=int(rng.integers(0, 3)))
        check_chunk(ctx, data, True)
        cut = int(rng.integers(0, len(data) + 1))
        check_chunk(ctx, data[:cut], False)


@pytest.mark.parametrize("seed", range(12))
def test_byte_soup_across_tile_edges(ctxs, seed):
    """Accidental structure only; long runs of spaces / bytes without '\\n' so that lines, space runs and ids cross the
    16 KiB tile edges in every state."""
    rng = np.random.default_rng(500 + seed)
    ctx = ctxs(bool(seed & 1), 32768 if seed % 3 == 0 else 256 * 1024)
    weights = [[2, 4, 3, 1, 6, 6, 1, 0.3, 0.3], [1, 0.02, 30, 5, 3, 3, 1, 0.05, 0.5], [0.5, 0.01, 1, 1, 20, 20, 0.2, 0.01, 0.1],
               [3, 0.3, 10, 2, 1, 1, 1, 0, 1]][seed % 4]
    for _ in range(6):
        n = int(rng.choice([100, TILE - 1, TILE, TILE + 1, 3 * TILE + 17, 100_000]))
        data = rand_soup(rng, n, weights)
        check_chunk(ctx, data, True)
        check_chunk(ctx, data, False)
        # a well-formed frame around the soup so that records exist on both sides of it
        framed = b">a\nAC\n>b " + data.replace(b">", b"A") + b"\n>c\nGT\n"
        check_chunk(ctx, framed, True)


def test_every_byte_value_is_classified_like_the_reference(ctxs):
    body = bytes(b for b in range(256) if b not in (10, 62))
    data = b">" + body + b"\n" + body + b"\n" + b"".join(bytes([b]) + b"\n" for b in range(256)) + b">z\n" + bytes(range(11, 256))
    for check in (False, True):
        check_chunk(ctxs(check), data, True)
    # without the bytes >= 0x80 so that the ascii check passes through to the end
    low = bytes(b for b in range(128) if b not in (10, 62))
    check_chunk(ctxs(True), b">" + low + b"\n" + low + b"\n>y\n" + low, True)


def test_spaces_and_states_straddling_tile_edges(ctxs):
    ctx = ctxs(False)
    for edge_fill in (b" ", b"\t", b"A"):
        for k in range(-3, 4):
            # header whose id / trailing spaces / '>' land on the tile edge
            pad = TILE - 8 + k
            check_chunk(ctx, b">a\n" + b"C" * (pad - 3 - 1) + b"\n" + b" > id  with  gaps   \n  AC  GT  \n" + edge_fill * 40 + b"\nTT\n", True)
            # sequence line that is all spaces for more than a tile, then an X or a '\n'
            for closer in (b"G\n", b"\n", b""):
                check_chunk(ctx, b">a\nAC" + edge_fill * (2 * TILE + k) + closer + b">b\nT\n", True)
                check_chunk(ctx, b">" + edge_fill * (TILE + k) + b"id" + b" " * (TILE + 5) + closer + b"ACGT\n", True)
                check_chunk(ctx, b" " * (TILE + k) + b">x" + b" " * TILE + closer + b"AC\n", True)
    # single-line records far longer than a tile (long reads as FASTA)
    rng = np.random.default_rng(5)
    recs = []
    for i in range(30):
        L = int(rng.integers(1, 120_000))
        recs.append(b">read%d\n" % i + np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)].tobytes() + b"\n")
    check_chunk(ctx, b"".join(recs), True)
    check_chunk(ctx, b"".join(recs), False)


def test_line_capacity(ctxs):   # buffered.mojo:634-636, 737-765
    cap = 32768
    ctx = ctxs(False, cap)
    ok = b">x\n" + b"A" * (cap - 1) + b"\n"
    assert check_chunk(ctx, ok, True).status == F.EOF
    for data in (b">x\nAC\n>y\n" + b"A" * cap + b"\nAC\n",               # inside record 1
                 b">x\nAC\n>" + b"y" * cap + b"\nAC\n",                  # a header line: record 0 is still open
                 b">x\n" + b"A" * cap,                                   # last line, no '\n'
                 b" " * cap + b"\n>x\nAC\n",                             # before any header
                 b"A" * (cap + 5) + b"\n>x\nAC\n",                       # too long wins over "no header"
                 b"AC\n" + b"A" * (cap + 5) + b"\n",                     # "no header" comes first
                 b">x\nAC\n>e\n>y\n" + b"A" * cap + b"\n",               # empty record 1 comes before the long line in record 2
                 b">x\nAC\n>e\n" + b" " * cap + b"\n>y\nA\n"):           # the long line is read while record 1 is open
        w = check_chunk(ctx, data, True)
        assert w.status != F.EOF
        check_chunk(ctx, data, False)
    assert check_chunk(ctx, b">x\n" + b"A" * (cap - 1), True).status == F.EOF
    # a header line that is too long: "too long" is met while the line is read, before it can close the record before it
    # (found by tests/fuzz_campaign_fasta.py: a chunk that ends inside that line must not close the record early)
    data = b"\r\n" * 50 + b">id\n" + b" >" * 20000 + b"\nAC\n"
    assert check_chunk(ctx, data, True).status == F.LINE_TOO_LONG
    for cut in (110, 128, 20000, len(data) - 4, len(data) - 3):
        check_chunk(ctx, data[:cut], False)


def test_streaming_parser_matches_whole_file_and_reference_tests(tmp_path):
    import blazeseq_amd as B
    rng = np.random.default_rng(77)
    data = rand_fasta(rng, 3000, 70, 6, dirty=0.0, crlf=False, lead_blank=2)
    want = F.flat_parse(data).records()
    for chunk in (1 << 12, 50_000, 1 << 26):
        p = B.FastaParser(data, chunk_bytes=chunk)
        got = [(r.id, r.sequence) for r in p.records()]
        assert got == want
        assert not p.has_more()
        with pytest.raises(B.ParseError):
            p.next_record()
        p.close()
    path = tmp_path / "x.fasta"
    path.write_bytes(data)
    p = B.FastaParser(str(path), chunk_bytes=100_000)
    assert [(r.id, r.sequence) for r in p] == want
    import gzip
    gz = tmp_path / "x.fasta.gz"
    gz.write_bytes(gzip.compress(data[:100_000], 1) + gzip.compress(data[100_000:], 6))   # two members, cut mid-record
    assert [(r.id, r.sequence) for r in B.FastaParser(str(gz), chunk_bytes=64_000)] == want
    # a record larger than the chunk makes the chunk grow
    big = b">big\n" + b"ACGT" * 50_000 + b"\n>small\nAC\n"
    assert [(r.id, len(r)) for r in B.FastaParser(big, chunk_bytes=4096).records()] == [(b"big", 200_000), (b"small", 2)]
    # errors: text and position are stream-global even when the error is chunks away from the start
    bad = data + b">empty\n>next\nAC\n"
    ref = F.flat_parse(bad)
    p = B.FastaParser(bad, chunk_bytes=30_000)
    n = 0
    with pytest.raises(B.ParseError) as e:
        while True:
            p.next_record()
            n += 1
    assert n == ref.n_records and e.value.message.decode("latin-1") == ref.message and e.value.code == ref.status
    # FastaRecord surface (tests/fasta/test_fasta_parser.mojo:835-1028)
    r = B.FastaRecord("id1 desc", "ACGTACGT")
    assert r.byte_len() == 1 + 8 + 1 + 8 + 1 and len(r) == 8 and r.write(4) == b">id1 desc\nACGT\nACGT\n"
    assert r == B.FastaRecord("other", "ACGTACGT") and r != B.FastaRecord("id1 desc", "ACGT")
    assert [(x.id, x.sequence) for x in B.FastaParser(r.write(3)).records()] == [(b"id1 desc", b"ACGTACGT")]


def test_device_generator_and_device_input(ctxs):
    import torch
    ctx = ctxs(False)
    want = F.generate_synthetic(3000, 5, 400, 60)
    t = ctx.generate_synthetic_device(3000, 5, 400, 60)
    assert bytes(t.cpu().numpy()) == want.tobytes()
    part = ctx.generate_synthetic_device(3000, 5, 400, 60, first=1234, count=500)
    whole = want.tobytes()
    idx = whole.index(b">read_1234\n")
    assert bytes(part.cpu().numpy()) == whole[idx:idx + part.numel()]
    res = ctx.parse(int(t.data_ptr()), t.numel(), True)   # device pointer: parsed in place
    w = F.flat_parse(whole)
    assert int(res.n_records) == 3000 == w.n_records and int(res.seq_bytes) == w.seq_bytes.size
    assert ctx.to_host(res.d_seq_bytes, int(res.seq_bytes), np.uint8).tobytes() == w.seq_bytes.tobytes()
    torch.cuda.synchronize()


def test_more_records_than_the_first_guess(ctxs):
    """Tiny records: far more headers than one per 64 bytes, so pass 2 is repeated with larger per-record arrays."""
    import blazeseq_amd as B
    ctx = B.FastaContext(B.FastaParserConfig(True))   # a fresh handle: nothing sized yet
    data = b"".join(b">%d\nA\n" % (i % 10) for i in range(200_000))
    w = check_chunk(ctx, data, True)
    assert w.n_records == 200_000
    check_chunk(ctx, data[:-1] + b"\n>\n", False)
    check_chunk(ctx, b">a\n" + b">\n" * 100_000, True)   # and an error far beyond the guess
    ctx.close()


def test_full_benchmark_size_properties(ctxs):
    """The reference's FASTA benchmark input at full size (1.5 M records of 200-3800 bp wrapped at 60 = 3.07 GB, generated on
    the device): every per-record number against its closed form, and windows of the columns against an independent
    parse of just those records (which the oracle checks byte for byte at this size in test_device_generator...)."""
    import torch
    ctx = ctxs(True)
    N, lo, hi, lw = 1_500_000, 200, 3800, 60
    t = ctx.generate_synthetic_device(N, lo, hi, lw)
    res = ctx.parse(int(t.data_ptr()), t.numel(), True)
    assert int(res.status) == F.EOF and int(res.n_records) == N
    i = np.arange(N, dtype=np.int64)
    L = lo + (i * 31 + 7) % (hi - lo + 1)
    seq_ends = ctx.to_host(res.d_seq_ends, N, np.int64)
    id_ends = ctx.to_host(res.d_id_ends, N, np.int64)
    hdr_pos = ctx.to_host(res.d_hdr_pos, N, np.int64)
    assert np.array_equal(seq_ends, np.cumsum(L))
    assert np.array_equal(id_ends, 12 * (i + 1))                      # "read_%07d"
    rec_bytes = 14 + L + (L + lw - 1) // lw                              # header line + bases + one '\n' per line
    assert np.array_equal(hdr_pos, np.cumsum(rec_bytes) - rec_bytes) and int(rec_bytes.sum()) == t.numel()
    assert int(res.seq_bytes) == int(L.sum()) and int(res.id_bytes) == 12 * N
    # newline-free, ACGT-only sequence column; id column = the headers without '>' and '\n'
    seq = _as_cuda(res.d_seq_bytes, int(res.seq_bytes))
    hist = torch.bincount(seq[: 1 << 28].to(torch.int64), minlength=256)
    assert int(hist.sum()) == int(hist[[65, 67, 71, 84]].sum())
    ctx2 = ctxs(False)
    for first in (0, 777_777, N - 1000):
        part = ctx2.generate_synthetic_device(N, lo, hi, lw, first=first, count=1000)
        r2 = ctx2.parse(int(part.data_ptr()), part.numel(), True)
        assert int(r2.n_records) == 1000
        s0 = int(seq_ends[first - 1]) if first else 0
        a = ctx.to_host(res.d_seq_bytes + s0, int(r2.seq_bytes), np.uint8)
        b = ctx2.to_host(r2.d_seq_bytes, int(r2.seq_bytes), np.uint8)
        assert np.array_equal(a, b)
        ia = ctx.to_host(res.d_id_bytes + 12 * first, 12000, np.uint8)
        assert ia.tobytes() == b"".join(b"read_%07d" % k for k in range(first, first + 1000))
    torch.cuda.synchronize()


def _as_cuda(ptr, nbytes):
    """A device pointer as a DLPack-able object (torch tensor view, no copy)."""
    import torch

    class _Holder:
        def __init__(self):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(_Holder(), device="cuda")


def test_giant_lines_do_not_make_the_tile_walks_quadratic(ctxs):
    """One line of hundreds of megabytes (a chromosome on one line, or a file of spaces): the reference refuses it
    ("Line exceeds buffer capacity"); here every tile walk is bounded by the line capacity, so it is also refused fast."""
    import time
    ctx = ctxs(False)
    big = 600 * 1000 * 1000
    for data, status in ((b">chr1\n" + b"A" * big + b"\n>next\nACGT\n", F.LINE_TOO_LONG),
                         (b">a\nAC\n>b\nGT\n" + b" " * big + b"\n>c\nA\n", F.LINE_TOO_LONG),
                         (b">a\nAC\n" + b" " * big + b">late\nA\n", F.LINE_TOO_LONG),
                         (b">a\nAC\n>b\n" + b"T" * big, F.LINE_TOO_LONG)):
        t0 = time.time()
        w = check_chunk(ctx, data, True)
        assert w.status == status
        check_chunk(ctx, data, False)   # as a chunk that is not the last one
        assert time.time() - t0 < 60, time.time() - t0
        del data


def _bgzf(data: bytes, block: int = 65280) -> bytes:
    """BGZF writer (SAM spec 4.1): independent gzip members with a 'BC' extra subfield + the empty EOF block."""
    import struct
    import zlib
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1)
                   + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


@pytest.mark.parametrize("kind", ["plain", "gzip", "bgzf"])
def test_files_through_the_native_ingest(kind, tmp_path):
    """A path goes through bzq_fasta_ingest_* (reader threads -> pinned double buffers -> device, carry on the device):
    same records, same terminal event and text as the oracle on the file's bytes, whatever the chunk size."""
    import gzip
    import blazeseq_amd as B
    rng = np.random.default_rng(31 + len(kind))
    for trial in range(4):
        data = rand_fasta(rng, int(rng.integers(2000, 6000)), int(rng.choice([60, 70, 200])), 6, dirty=[0.0, 0.0, 0.01, 0.0][trial],
                          crlf=bool(trial & 1), tail_newline=trial != 3, lead_blank=trial)
        if trial == 2:
            data += b">empty\n>after\nAC\n"
        comp = data if kind == "plain" else (gzip.compress(data[: len(data) // 3], 1) + gzip.compress(data[len(data) // 3:], 6) if kind == "gzip"
                                             else _bgzf(data))
        path = tmp_path / ("t%d.fa%s" % (trial, "" if kind == "plain" else ".gz"))
        path.write_bytes(comp)
        want = F.flat_parse(data, True)
        for chunk in (1 << 16, 200_000, 1 << 26):
            p = B.FastaParser(str(path), B.FastaParserConfig(True), chunk_bytes=chunk, reader_threads=3)
            assert p._ingest is not None
            got, err = [], None
            try:
                while True:
                    r = p.next_record()
                    got.append((r.id, r.sequence))
            except B.ParseError as e:
                err = e
            assert got == want.records(), (kind, trial, chunk, len(got), want.n_records)
            assert err.code == want.status
            if want.status != F.EOF:
                assert err.message.decode("latin-1") == want.message
            st = p._ingest.stats()
            assert st.records == want.n_records and st.file_bytes == len(comp)
            p.close()
    # a record larger than a chunk: the open record is carried on the device until its end arrives
    big = b">big\n" + (b"ACGT" * 15 + b"\n") * 6667 + b">small\nAC\n"
    path = tmp_path / "big.fa"
    path.write_bytes(big)
    assert [(r.id, len(r)) for r in B.FastaParser(str(path), chunk_bytes=1 << 16).records()] == [(b"big", 60 * 6667), (b"small", 2)]
    with pytest.raises(RuntimeError, match="cannot open"):
        B.FastaIngest(B.FastaContext(), str(tmp_path / "nope.fa"))


def test_committed_golden_vectors(ctxs):
    """tests/golden/fasta_expected.json through the C ABI: digest of all five outputs, terminal status and text."""
    import hashlib
    import json
    import importlib.util
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("make_golden_fasta", os.path.join(here, "golden", "make_golden_fasta.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = json.load(open(os.path.join(here, "golden", "fasta_expected.json")))
    for key, want in gold.items():
        kind, name = key.split(":", 1)
        data = (open(os.path.join(GOLD, name), "rb").read() if kind == "file" else mg.CONSTRUCTED[name] if kind == "stream"
                else F.generate_synthetic(2000, 5, 400, 60).tobytes())
        for cfg, check in (("plain", False), ("check_ascii", True)):
            if cfg not in want:
                continue
            ctx = ctxs(check)
            res = ctx.parse(data, len(data), True)
            idb, ide, sqb, sqe, hp = ctx.columns(res)
            h = hashlib.sha256()
            for a in (idb, ide, sqb, sqe, hp):
                h.update(a.tobytes())
            w = want[cfg]
            assert (int(res.n_records), int(res.status), h.hexdigest()) == (w["n_records"], w["status"], w["digest"]), (key, cfg)
            if w["status"] != F.EOF:
                assert ctx.error_text().decode("latin-1") == w["message"]


def test_sequence_column_feeds_the_device_consumers(ctxs):
    """The FASTA columns are ordinary device arrays: the byte-histogram consumer of the FASTQ side (bzq_column_histogram,
    SURVEY 8f rank 2: GC / base counts) runs on the sequence column where it lies, no host round trip."""
    import ctypes as C
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    ctx = ctxs(False)
    data = F.generate_synthetic(20_000, 50, 900, 60).tobytes()
    res = ctx.parse(data, len(data), True)
    want = F.flat_parse(data)
    fq = B.Context(B.ParserConfig(), "generic", 4096, 0)
    hist = (C.c_uint64 * 256)()
    assert L.lib().bzq_column_histogram(fq.h, C.c_void_p(res.d_seq_bytes), int(res.seq_bytes), hist) == 0
    assert np.array_equal(np.frombuffer(hist, dtype=np.uint64).astype(np.int64), np.bincount(want.seq_bytes, minlength=256))
    gc = (hist[ord("G")] + hist[ord("C")]) / int(res.seq_bytes)
    assert 0.45 < gc < 0.55   # gc_bias = 0.5 in the generator
    fq.close()


def test_reference_record_tests_through_the_gpu_parser():
    """tests/fasta/test_fasta_parser.mojo:835-1028 with the records coming from the GPU parser: byte_len / len / write /
    read -> write -> read fidelity (the host-only half is tests/test_fasta_record_host.py)."""
    import blazeseq_amd as B

    def records(data):
        p = B.FastaParser(data)
        out = list(p.records())
        p.close()
        return out
    assert records(b">abc\nACGT\n")[0].byte_len() == 10
    assert len(records(b">id1\nACGT\nACGT\n")[0]) == 8
    assert records(b">id1\nACGT\n")[0].write() == b">id1\nACGT\n"
    assert records(b">id\nAC\nGT\n")[0].byte_len() == 9
    r = records(b">myid\nGATTACA\n")[0]
    assert (r.id, r.sequence, len(r), r.byte_len()) == (b"myid", b"GATTACA", 7, 14)
    for data in (b">id1\nACGT\n", b">id1\nACGT\n>id2\nTTAA\n>id3\nGGCC\n", b">id1 description here\nACGT\n", b">seq1\nACG\nTTA\nGG\n",
                 b">long\n" + b"ACGT" * 100 + b"\n"):
        original = records(data)
        again = records(b"".join(x.write() for x in original))
        assert [(x.id, x.sequence) for x in original] == [(x.id, x.sequence) for x in again] and len(original) >= 1
    assert records(b">id1 description here\nACGT\n")[0].id == b"id1 description here"
    # has_more / exhaustion (:289, :307)
    p = B.FastaParser(b">id1\nACGT\n")
    assert p.has_more()
    p.next_record()
    assert not p.has_more()
    with pytest.raises(B.ParseError) as e:
        p.next_record()
    assert e.value.code == F.EOF
    p.close()
