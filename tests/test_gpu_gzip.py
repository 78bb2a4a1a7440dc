"""bzq_gzip_* (blazeseq_amd/csrc/bzq_gzip.hpp): ANY gzip stream inflated on the GPU in parallel == the bytes zlib gives.
The checker is zlib itself -- the library behind the reference's GZFile (blazeseq/io/readers.mojo:226-377), and what
rapidgzip's output (RapidgzipReader, readers.mojo:380-443) is defined to equal."""
import gzip
import os
import zlib

import numpy as np
import pytest

from blazeseq_amd.parser import Context
from tests.fastq_fuzz import rand_stream
from tests.gzip_util import DeviceGunzip, gzip_member, sequencer_like

pytestmark = pytest.mark.gpu


def synthetic_fastq(n_records: int, seed: int = 5) -> bytes:
    from oracle import oracle as O
    return O.generate_synthetic(n_records, 150, 150, 33, 73, "generic").tobytes()


def payloads():
    rng = np.random.default_rng(21)
    yield "fastq_small", rand_stream(rng, n_records=3000, max_len=150, dirty=0.0, tail=0)
    yield "fastq_4mb", synthetic_fastq(13000)
    yield "empty", b""
    yield "one_byte", b"A"
    yield "same_byte", b"G" * 700000
    yield "period3", b"ACG" * 200000
    yield "random", rng.integers(0, 256, 400000, dtype=np.uint8).tobytes()
    yield "low_entropy", rng.integers(65, 69, 900000, dtype=np.uint8).tobytes()
    yield "text", (b"the quick brown fox jumps over the lazy dog\n" * 20000)[:800001]
    yield "far_matches", (bytes(rng.integers(0, 256, 31000, dtype=np.uint8)) * 20)


FINDABLE_PAYLOADS = {"fastq_small", "fastq_4mb", "text", "low_entropy"}   # compressible: zlib writes dynamic-Huffman blocks for them


@pytest.mark.parametrize("host_continuation", [0, 1])
@pytest.mark.parametrize("name,data", list(payloads()), ids=[n for n, _ in payloads()])
def test_levels_and_strategies(name, data, host_continuation):
    """Every level / strategy of zlib against zlib's own output, with the device ALONE (host_continuation = 0: no byte of the result
    comes from the checker's library) and with the default.  With the default, a stream whose blocks the finder can start in
    (dynamic Huffman) must not have touched the host either: `host_calls` is read for every row."""
    ctx = Context()
    for level, strategy in [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
                            (6, zlib.Z_FIXED), (6, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_FILTERED)]:
        comp = gzip_member(data, level, strategy)
        g2 = DeviceGunzip(ctx, len(data) + 4096, chunk_bytes=4096)
        g2.dec.set_option("host_continuation", host_continuation)
        try:
            got = g2.decode(comp)
            st = g2.dec.stats()
            host_calls = g2.dec.set_option("host_calls", 0)
        finally:
            g2.close()
        assert got == data, (name, level, strategy, len(got), len(data))
        assert st.members == 1 and st.bytes_out == len(data)
        findable = level != 0 and strategy != zlib.Z_FIXED and name in FINDABLE_PAYLOADS
        if host_continuation == 0 or findable:
            assert host_calls == 0, (name, level, strategy, host_calls)


def test_the_speculation_is_what_runs():
    """On an ordinary file the chunks find their own starts and the chain runs through them: many decoder waves end up in the
    output, nothing is restarted."""
    data = synthetic_fastq(40000)   # 12.7 MB
    comp = gzip.compress(data, 6)
    ctx = Context()
    g = DeviceGunzip(ctx, len(data) + 4096)
    assert g.decode(comp) == data
    st = g.dec.stats()
    g.close()
    assert st.chain_jobs >= 20 and st.fallback_jobs == 0, (st.chain_jobs, st.fallback_jobs)
    assert st.chunks_with_start >= st.chain_jobs - 1


def test_header_fields_and_members():
    rng = np.random.default_rng(3)
    parts = [synthetic_fastq(2000), b"", rand_stream(rng, n_records=500, max_len=90, dirty=0.0, tail=0), b"x", synthetic_fastq(5000)]
    comp = b"".join([
        gzip_member(parts[0], 6, name=b"reads_1.fastq"),
        gzip_member(parts[1], 6),
        gzip_member(parts[2], 9, name=b"n" * 300, comment=b"a comment", extra=b"AB\x04\x00abcd", hcrc=True),
        gzip_member(parts[3], 1),
        gzip_member(parts[4], 6, flush_every=50000),
    ])
    want = b"".join(parts)
    assert gzip.decompress(comp) == want
    ctx = Context()
    for host_continuation in (0, 1):
        for piece in (0, 1 << 16, 70001, 999):
            g = DeviceGunzip(ctx, len(want) + 4096, chunk_bytes=4096)
            g.dec.set_option("host_continuation", host_continuation)
            got = g.decode(comp, piece)
            st = g.dec.stats()
            host_calls = g.dec.set_option("host_calls", 0)
            g.close()
            assert got == want, piece
            assert st.members == 5 and g.dec.finished is not None
            assert host_calls == 0, (host_continuation, piece, host_calls)   # (every member here has dynamic blocks or is tiny)


def test_many_small_members_decode_in_parallel():
    """A file of many small members that is not BGZF (no BC field): member headers are block starts like any other."""
    data = synthetic_fastq(30000)
    size = 60000
    comp = b"".join(gzip_member(data[i:i + size], 6) for i in range(0, len(data), size))
    ctx = Context()
    g = DeviceGunzip(ctx, len(data) + 4096, chunk_bytes=8192)
    assert g.decode(comp) == data
    st = g.dec.stats()
    g.close()
    assert st.members == (len(data) + size - 1) // size
    assert st.chain_jobs >= st.members // 4 and st.fallback_jobs == 0


def test_pieces_and_small_output_buffers():
    """The stream in pieces of every size (cuts inside headers, blocks, trailers), and an output buffer far smaller than the
    output: the decoder hands over what fits and keeps the rest."""
    data = synthetic_fastq(20000)
    comp = gzip.compress(data, 6)
    ctx = Context()
    for piece, cap in [(1 << 20, 1 << 20), (100000, len(data)), (12345, 600000), (len(comp) - 3, len(data)), (len(comp) - 8, len(data))]:
        g = DeviceGunzip(ctx, cap, chunk_bytes=4096)
        assert g.decode(comp, piece) == data, (piece, cap)
        g.close()


def test_staged_pieces_are_the_pieces():
    """bzq_gzip_stage: pieces sent to the device ahead of their decode call (one or two in front, the second refused when both
    buffers are taken), with an output buffer that cuts pieces (decode calls without input in between), and a staged piece
    that is never fed (dropped): the bytes are the same."""
    data = synthetic_fastq(30000)
    comp = gzip.compress(data, 6)
    ctx = Context()
    for piece, cap, ahead in [(1 << 20, len(data), 1), (300000, len(data), 2), (99991, 700000, 1), (50000, 400000, 3), (len(comp), len(data), 1)]:
        g = DeviceGunzip(ctx, cap, chunk_bytes=4096)
        assert g.decode(comp, piece, ahead=ahead) == data, (piece, cap, ahead)
        g.close()
    g = DeviceGunzip(ctx, len(data), chunk_bytes=4096)
    stray = np.frombuffer(comp, dtype=np.uint8)[1000:50000].copy()
    g.dec.stage(stray)
    g.dec.stage(stray[:100])
    assert g.decode(comp, 200000, ahead=1) == data
    g.close()


def test_what_a_sequencer_writes():
    """Quality values in long runs (matches that overlap themselves: distance 1, length up to 258), duplicate reads (matches as
    long as a read), poly-G tails: the symbol loop takes such matches in pieces, lane i of a piece reading source symbol
    i mod distance."""
    data, _ = sequencer_like(6 << 20)
    ctx = Context()
    for level in (1, 6, 9):
        g = DeviceGunzip(ctx, len(data) + 4096)
        assert g.decode(gzip.compress(data, level)) == data, level
        g.close()


def test_trailing_garbage_is_ignored_like_gzread_does():
    data = synthetic_fastq(3000)
    comp = gzip.compress(data, 6)
    ctx = Context()
    for junk in (b"\0" * 100, b"some text that is no gzip member", b"\x1f", b"\x1f\x00\x8b"):
        g = DeviceGunzip(ctx, len(data) + 4096)
        assert g.decode(comp + junk, 50000) == data
        assert g.dec.finished
        g.close()


def test_a_later_member_that_is_cut_off_is_a_truncated_file():
    """once the magic 1f 8b has matched, gzread reports "unexpected end of file" for a member cut inside its header
    or right behind it -- trailing bytes are only ignored when they are no member at all."""
    data = synthetic_fastq(3000)
    comp = gzip.compress(data, 6)
    second = gzip.compress(b"@r\nACGT\n+\nIIII\n", 6)
    ctx = Context()
    for cut in (2, 3, 9, 10, 11, len(second) - 9):   # the magic alone, inside the fixed header, right behind it, inside the block, no trailer
        g = DeviceGunzip(ctx, len(data) + 4096)
        with pytest.raises(RuntimeError, match="truncated"):
            g.decode(comp + second[:cut], 50000)
        g.close()
        with pytest.raises((EOFError, Exception)):
            gzip.decompress(comp + second[:cut])   # (zlib agrees)


def test_corrupt_streams_fail_the_call():
    data = synthetic_fastq(8000)
    comp = bytearray(gzip.compress(data, 6))
    ctx = Context()
    # a flipped bit in the middle of the DEFLATE data, a wrong CRC, a wrong ISIZE, a truncated file
    cases = {"bit_flip": bytearray(comp), "crc": bytearray(comp), "isize": bytearray(comp), "truncated": bytearray(comp[:len(comp) // 2])}
    cases["bit_flip"][len(comp) // 2] ^= 0x10
    cases["crc"][-6] ^= 0xFF
    cases["isize"][-2] ^= 0x01
    for name, bad in cases.items():
        g = DeviceGunzip(ctx, len(data) + 4096, chunk_bytes=4096)
        with pytest.raises(RuntimeError):
            g.decode(bytes(bad), 200000)
        g.close()
    g = DeviceGunzip(ctx, 4096)
    with pytest.raises(RuntimeError):
        g.decode(b"this is not gzip at all, not even close" * 10)
    g.close()


def test_random_streams_against_zlib():
    """300 random streams: content kinds x levels x strategies x memLevel x member splits x piece sizes x chunk sizes."""
    rng = np.random.default_rng(77)
    ctx = Context()
    for it in range(300):
        kind = it % 6
        n = int(rng.integers(0, 400000))
        if kind == 0:
            data = rand_stream(rng, n_records=int(rng.integers(1, 3000)), max_len=int(rng.integers(10, 300)), dirty=0.0, tail=0)
        elif kind == 1:
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 2:
            data = rng.integers(65, 65 + int(rng.integers(1, 20)), n, dtype=np.uint8).tobytes()
        elif kind == 3:
            unit = rng.integers(0, 256, int(rng.integers(1, 40000)), dtype=np.uint8).tobytes()
            data = (unit * (n // max(1, len(unit)) + 1))[:n]
        elif kind == 4:
            data = bytes(rng.integers(0, 4, n, dtype=np.uint8) * 17 + 40)
        else:
            data = synthetic_fastq(int(rng.integers(1, 2500)))
        n_members = int(rng.integers(1, 5))
        cuts = sorted(int(x) for x in rng.integers(0, len(data) + 1, n_members - 1))
        parts = [data[a:b] for a, b in zip([0] + cuts, cuts + [len(data)])]
        comp = b"".join(gzip_member(p, int(rng.integers(0, 10)), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])),
                                    mem_level=int(rng.integers(1, 10)), flush_every=int(rng.choice([0, 0, 30000]))) for p in parts)
        piece = int(rng.choice([0, 1 << 16, int(rng.integers(1, len(comp) + 2))]))
        g = DeviceGunzip(ctx, len(data) + 4096, chunk_bytes=int(rng.choice([4096, 8192, 32768])))
        got = g.decode(comp, piece)
        g.close()
        assert got == data, (it, kind, len(data), n_members, piece)


# ---- round 4: a stretch without block starts the finder can find is continued on the host (gz_decode_host) -----------------------
def _unfindable(data: bytes, kind: str) -> bytes:
    """One member whose DEFLATE blocks the finder cannot recognise: fixed-Huffman blocks only (Z_FIXED) / stored blocks only
    (level 0).  On the device such a stream is ONE wave's work (~10 MB/s): the decoder stops (ST_FAR) and zlib goes on."""
    return gzip_member(data, 6, zlib.Z_FIXED) if kind == "fixed" else gzip_member(data, 0)


@pytest.mark.parametrize("kind", ["fixed", "stored"])
def test_throughput_floor_of_a_stream_without_findable_block_starts(kind):
    """'a stream of fixed-Huffman or stored blocks degrades to one wave.  No test pins the throughput floor there.'
    32 MB of FASTQ: 2.6 s (12 MB/s) on one wave; with the host continuation 0.14 s (230 MB/s, fixed) / 0.06 s (stored) on the
    GPU box's host.  The floor asserted is far below that (a loaded host) and far above one wave."""
    import time
    data = synthetic_fastq(100_000)
    comp = _unfindable(data, kind)
    ctx = Context()
    best = None
    for _ in range(2):
        g = DeviceGunzip(ctx, len(data) + (1 << 20))
        t0 = time.perf_counter()
        out = g.decode(comp)
        dt = time.perf_counter() - t0
        assert out == data
        assert g.dec.set_option("host_calls", 0) >= 1
        g.close()
        best = dt if best is None else min(best, dt)
    assert len(data) / best / 1e6 >= 60.0, f"{kind}: {len(data) / best / 1e6:.1f} MB/s"


def test_host_continuation_emits_more_than_one_pinned_block_per_call():
    """the host continuation's output buffer is lazily pinned memory registered in 32 MiB blocks, and one copy must not
    span two registrations.  Under the DEFAULT options (budget 32 MiB + 4 MiB of room, doubling while the device keeps handing over)
    a long stretch without findable block starts makes one call emit more than a block: 120 MB of stored blocks in one member."""
    data = synthetic_fastq(380_000)   # ~120 MB
    comp = _unfindable(data, "stored")
    ctx = Context()
    g = DeviceGunzip(ctx, len(data) + (1 << 20))
    out = g.decode(comp)
    assert len(out) == len(data) and out == data
    calls, mib = g.dec.set_option("host_calls", 0), g.dec.set_option("host_out_mib", 0)
    assert calls >= 1 and mib >= 100 and calls * 32 <= mib, (calls, mib)   # nearly all of it by the host, >= 32 MiB (+ the block that crosses the budget) per call: every call's copy spans two or more 32 MiB registrations
    g.close()


@pytest.mark.parametrize("kind", ["fixed", "stored"])
def test_device_and_host_hand_the_stream_back_and_forth(kind):
    """A small host budget: device (until ST_FAR) -> host (budget) -> device -> ... inside ONE member -- every hand-over passes the
    32 KiB window, the bit position inside a byte, the running CRC-32 and length -- through pieces of several sizes and an
    output buffer smaller than the output."""
    data = synthetic_fastq(26_000)   # 8 MB
    comp = _unfindable(data, kind)
    ctx = Context()
    for piece, cap, budget_kib, far_kib in [(0, len(data), 256, 64), (300_000, len(data), 64, 16), (1 << 20, 700_000, 512, 32), (77_777, 3 << 20, 100, 16)]:
        g = DeviceGunzip(ctx, cap, chunk_bytes=4096)
        g.dec.set_option("host_budget_kib", budget_kib)
        g.dec.set_option("far_kib", far_kib)
        assert g.decode(comp, piece) == data, (kind, piece, cap, budget_kib)
        assert g.dec.set_option("host_calls", 0) >= 3
        g.close()
    # and without the continuation the device does it alone (one wave: a small stream)
    g = DeviceGunzip(ctx, 1 << 20)
    g.dec.set_option("host_continuation", 0)
    assert g.decode(_unfindable(data[:600_000], kind)) == data[:600_000]
    assert g.dec.set_option("host_calls", 0) == 0
    g.close()


def test_host_continuation_across_members_trailers_and_the_end_of_the_stream():
    """Members of every kind in one file -- findable (dynamic) and not (fixed, stored), small and large, with header fields --
    so that the host meets member trailers, the next member's header, a member it hands back to the device, the last
    member's end, and bytes behind it that are no member (ignored, like gzread)."""
    rng = np.random.default_rng(77)
    parts, want = [], []
    for i in range(14):
        n = int(rng.choice([0, 1, 500, 70_000, 700_000, 2_000_000]))
        piece = synthetic_fastq(max(1, n // 318 + 1), seed=i)[:n]
        kind = i % 4
        m = (gzip_member(piece, 6, zlib.Z_FIXED, name=b"f%d" % i) if kind == 0 else gzip_member(piece, 0, comment=b"stored") if kind == 1
             else gzip_member(piece, 6) if kind == 2 else gzip_member(piece, 1, zlib.Z_FIXED, extra=b"xx\x02\x00ab", hcrc=True))
        parts.append(m); want.append(piece)
    comp, data = b"".join(parts), b"".join(want)
    ctx = Context()
    for piece, cap, tail in [(0, len(data) + 1, b""), (200_000, len(data) + 1, b"\0" * 300), (33_333, 900_000, b"not a member")]:
        g = DeviceGunzip(ctx, cap, chunk_bytes=4096)
        g.dec.set_option("host_budget_kib", 128)
        g.dec.set_option("far_kib", 32)
        assert g.decode(comp + tail, piece) == data, (piece, cap)
        assert g.dec.stats().members == 14 and g.dec.set_option("host_calls", 0) >= 2
        g.close()


def test_damage_met_by_the_host_fails_the_call():
    """A truncated file, a wrong CRC-32, a wrong ISIZE and invalid DEFLATE data inside a stretch the host decodes: a runtime
    error, never bytes."""
    data = synthetic_fastq(13_000)   # 4 MB
    good = _unfindable(data, "fixed")
    ctx = Context()

    def run(comp):
        g = DeviceGunzip(ctx, len(data) + 1, chunk_bytes=4096)
        g.dec.set_option("far_kib", 16)
        try:
            return g.decode(comp, 500_000)
        finally:
            g.close()

    assert run(good) == data
    for cut in (len(good) - 1, len(good) - 8, len(good) - 9, len(good) // 2):
        with pytest.raises(RuntimeError, match="truncated|unexpected end"):
            run(good[:cut])
    for off, what in ((-8, "CRC-32"), (-4, "length")):
        bad = bytearray(good); bad[off] ^= 0x40
        with pytest.raises(RuntimeError, match=what):
            run(bytes(bad))
    rng = np.random.default_rng(3)
    refused = 0
    for _ in range(12):
        bad = bytearray(good)
        for _ in range(3):
            bad[int(rng.integers(len(good) // 3, len(good) - 8))] ^= 1 << int(rng.integers(0, 8))
        try:
            assert run(bytes(bad)) == data   # (cannot happen: the CRC-32 would have to agree)
        except RuntimeError as e:
            assert "invalid DEFLATE data" in str(e) or "CRC-32" in str(e) or "length" in str(e) or "truncated" in str(e) or "unexpected end" in str(e), str(e)
            refused += 1
    assert refused == 12


def test_a_file_without_findable_block_starts_through_the_parser(tmp_path):
    """The whole way: FastqParser(path) on a fixed-Huffman-only .gz == the streaming oracle on the plain bytes."""
    import blazeseq_amd as B
    from oracle import oracle as O
    data = bytes(O.generate_synthetic(30_000, 50, 150, 0, 40, "sanger"))
    path = tmp_path / "fixed_only.fastq.gz"
    path.write_bytes(gzip_member(data, 6, zlib.Z_FIXED))
    ref = [b for b in O.StreamParser(np.frombuffer(data, dtype=np.uint8), O.make_config(batch_size=1000)).batches()]
    for chunk in (1 << 18, 1 << 22):
        got = list(B.FastqParser(str(path), batch_size=1000, chunk_bytes=chunk, reader_threads=2).batches())
        assert [len(b) for b in got] == [len(b) for b in ref]
        for g, r in zip(got, ref):
            assert g._sequence_bytes.tobytes() == r.seq_bytes and g._quality_bytes.tobytes() == r.qual_bytes and g._id_bytes.tobytes() == r.id_bytes


@pytest.mark.parametrize("early_find,predecode,chain_l2", [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0), (0, 1, 1), (1, 1, 1)])
def test_the_next_piece_under_this_one(early_find, predecode, chain_l2):
    """Round 4: with pieces staged ahead, piece k + 1's decoders run under piece k's chain / resolve / CRC kernels (option
    predecode, into the second set of pool / results), and its finder either behind the decoding of piece k (on the piece's
    uniform chunk grid) or behind its own copy (option early_find: on a grid over its own bytes, shifted once the carry is
    known).  Round 5: option chain_l2 -- beside a predecode the chain kernels run in the form that fits what the decoders leave of
    a CU (k_gz_chainl_*: windows through the L2).  Every combination == zlib's bytes, through piece sizes that cut blocks, headers
    and trailers, output buffers that cut pieces, many small members, and a stretch the host continues."""
    rng = np.random.default_rng(31)
    fq = synthetic_fastq(60_000)   # 19 MB
    many = b"".join(gzip_member(fq[i:i + 150_000], int(rng.integers(1, 10))) for i in range(0, 6_000_000, 150_000))
    mixed = gzip_member(fq[:5_000_000], 6) + gzip_member(fq[5_000_000:7_000_000], 6, zlib.Z_FIXED) + gzip_member(fq[7_000_000:], 9)
    ctx = Context()
    for comp, data, piece, cap, ahead in [(gzip.compress(fq, 6), fq, 1 << 20, len(fq), 2), (gzip.compress(fq, 1), fq, 700_001, 5_000_000, 2),
                                          (many, fq[:6_000_000], 400_000, 6_000_000, 2), (mixed, fq, 1_500_000, len(fq), 2),
                                          (gzip.compress(fq, 6), fq, 2_000_000, len(fq), 1)]:
        g = DeviceGunzip(ctx, cap, chunk_bytes=4096)
        g.dec.set_option("early_find", early_find)
        g.dec.set_option("predecode", predecode)
        g.dec.set_option("chain_l2", chain_l2)
        g.dec.set_option("far_kib", 64)
        assert g.decode(comp, piece, ahead=ahead) == data, (early_find, predecode, chain_l2, piece, cap, ahead)
        g.close()


def test_device_memory_a_gz_stream_holds():
    """'bound the gzip pool ... with a test that asserts the hipMemGetInfo delta'.  A .gz stream's device memory scales with
    chunk_bytes: two symbol pools (pages for the piece's output at the stream's own ratio + one page per decoder job: ~13 x the
    compressed piece each, 3.3 GiB at a 256 MiB piece), ONE output FIFO of 6 x chunk_bytes, three slots, three buffers of compressed
    bytes, the parser's own arenas: **3.5 GiB at 64 MiB chunks, 10.6 GiB at the default 256 MiB** (rounds 3-5: 5.6 / 13.2 and more -- pools
    reserved for a ratio of 5 and two pages per job, and a second FIFO that every chunk's remainder moved to).  The pools alone are
    now below the 4 GiB VERDICT asked for at 128 MiB pieces and not at 256 MiB (smaller pieces cost rate,
    profiles/r5_gzip_piece_sweep.txt); what is pinned here is that the footprint is what INTEGRATION.md says and does not double
    when a piece overflows the pool (round 4: 4 GiB -> 8 GiB for good).  Measured with the library's buffer cache off, after the whole
    file has been decoded."""
    import ctypes as C
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    L.lib()
    # (the HIP runtime the library is linked against, already in the process -- not torch: importing torch AFTER another copy of the
    # runtime has been initialised in the process does not return)
    hip = C.CDLL("libamdhip64.so.7")

    def mem_free():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    # 1.7 GB of FASTQ, 0.8 GB compressed: four pieces at the default chunk size, so that BOTH pools have seen a whole-chunk piece (a
    # shorter file ends before the first pool, sized for the stream's first half-chunk piece, has grown: 8.6 GiB instead of 10.1)
    n_member, reps = 100_000, 54
    member = gzip_member(synthetic_fastq(n_member), 6)
    n_total = n_member * reps
    path = "/dev/shm/bzq_footprint_test.fastq.gz" if os.path.isdir("/dev/shm") else "/tmp/bzq_footprint_test.fastq.gz"
    with open(path, "wb") as f:
        for _ in range(reps):
            f.write(member)
    try:
        ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
        for key in ("pin_cache_bytes", "dev_cache_bytes"):
            ctx.set_option(key, 0)
        # measured 3.48 / 10.58 GiB (round 5, after the pool / FIFO diet; 5.62 / 13.24 on a file of two pieces before), parser arenas included --
        # and 12.25 GiB at 256 MiB when piece 1 is planned before piece 0's ratio is known (its finder and decoders run under piece 0's
        # chain: which one happens first is timing), so that BOTH pools are sized by the first-piece rule (3 x instead of the stream's
        # own 2.1 x 1.25): the bound below is that case, the deterministic upper end (round 6: seen once in four runs)
        for chunk_mib, limit_gib in ((64, 4.6), (256, 12.6)):
            assert hip.hipDeviceSynchronize() == 0
            free0 = mem_free()
            ing = B.Ingest(ctx, path, chunk_bytes=chunk_mib << 20, n_threads=4)
            taken = total = 0
            low = free0
            while True:
                r = ing.next(taken)
                taken = int(r.n_records); total += taken
                low = min(low, mem_free())
                if int(r.status) != L.OK:
                    break
            assert total == n_total and int(r.status) == L.EOF
            held = (free0 - low) / 2**30
            ing.close()
            assert hip.hipDeviceSynchronize() == 0
            print(f"chunk {chunk_mib} MiB: {held:.2f} GiB")
            assert held <= limit_gib, f"chunk {chunk_mib} MiB: the stream held {held:.2f} GiB of device memory (limit {limit_gib})"
        for key in ("pin_cache_bytes", "dev_cache_bytes"):
            ctx.set_option(key, 1 << 30)
        ctx.close()
    finally:
        os.remove(path)
