"""GPU parity: the HIP path (through the C ABI) must be bit-identical to the CPU oracle -- record
offsets, lengths, validation outcome, error text, FastqBatch columns -- on the reference's corpus,
on adversarial streams (ragged, truncated, CRLF, non-ASCII, empty), with every kernel path (fast /
serial / multi-pass) and every ParserConfig switch."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from fastq_fuzz import rand_stream
from blazeseq_amd import _lib as L_
from gpu_util import make_pair, check_against_oracle, EXPERIMENTS, VARIANTS, VARIANTS_LB, SHARD_VARIANTS

HERE = os.path.dirname(os.path.abspath(__file__))
CORPUS = json.load(open(os.path.join(HERE, "golden", "corpus_expected.json")))


@pytest.mark.parametrize("single_pass", VARIANTS)
def test_inline_known_answers(single_pass):
    ctx, oc = make_pair(emit_offsets=True, single_pass=single_pass)
    for data in (b"@r1\nACGT\n+\n!!!!\n@r2\nTGCA\n+\n####\n", b"", b"\n", b"@", b"@a\nA\n+\n!\n",
                 b"r1\nACGT\n+\n!!!!\n", b"@r1\nACGT\n+\n!!!\n", b"@r1\nACGT\n-\n!!!!\n",
                 b"@a\nAC\n+\n!!\n@b\nACGT\n+\n!!", b"@a\nAC\n+\n!!\n@b\nAC\n+\n \t", b"@a\nAC\n+\n!!\n\n",
                 b"@only\nACGT", b"@a\nAC\n+\n!!\n@only\nACGT", b"@id\r\nACGT\r\n+\r\n!!!!\r\n",
                 b"@ \t id with spaces \t\nAC\n+\n!!\n@\nAC\n+\n!!\n@   \nAC\n+\n!!\n", b"\n\n\n\n", b"\n\n\n\n\n\n\n\n\n"):
        check_against_oracle(ctx, oc, data, offsets=True, what=repr(data[:20]))


@pytest.mark.parametrize("single_pass", VARIANTS)
@pytest.mark.parametrize("cfgname", ["default", "validated_generic", "validated_schema", "validated_schema_simd32",
                                     "cap64", "cap64_growth"])
def test_corpus(cfgname, single_pass, corpus_dir):
    for name, e in sorted(CORPUS.items()):
        data = open(os.path.join(corpus_dir, name), "rb").read()
        sc = e["schema"]
        kw = {"default": {}, "validated_generic": dict(check_ascii=True, check_quality=True),
              "validated_schema": dict(check_ascii=True, check_quality=True, quality_schema=sc),
              "validated_schema_simd32": dict(check_ascii=True, check_quality=True, quality_schema=sc, compat_simd_width=32),
              "cap64": dict(buffer_capacity=64),
              "cap64_growth": dict(buffer_capacity=64, buffer_growth_enabled=True, buffer_max_capacity=1 << 20)}[cfgname]
        ctx, oc = make_pair(emit_offsets=True, single_pass=single_pass, **kw)
        res, f = check_against_oracle(ctx, oc, data, offsets=True, what=f"{name}/{cfgname}")
        g = e[cfgname]
        assert (int(res.n_records), res.status, ctx.format_error().decode("latin-1") if res.status else "") == \
               (g["n_records"], g["term_code"], g["term_msg"]), name
        ctx.close()


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_small(seed):
    rng = np.random.default_rng(1000 + seed)
    for rep in range(6):
        data = rand_stream(rng, n_records=int(rng.integers(0, 60)), max_len=int(rng.integers(1, 80)),
                           dirty=float(rng.choice([0.0, 0.02, 0.1])), crlf=bool(rng.random() < 0.15))
        for kw in (dict(), dict(check_ascii=True, check_quality=True),
                   dict(check_ascii=True, check_quality=True, quality_schema="solexa", compat_simd_width=16),
                   dict(buffer_capacity=48), dict(buffer_capacity=48, buffer_growth_enabled=True, buffer_max_capacity=200)):
            ctx, oc = make_pair(batch_size=int(rng.choice([1, 3, 4096])), emit_offsets=True, single_pass=VARIANTS[rep % len(VARIANTS)], **kw)
            check_against_oracle(ctx, oc, data, offsets=True, what=f"seed{seed}/{rep}/{kw}")
            if rep == 0:
                ctx.set_option("force_dense", 1)   # every tile through the serial in-kernel path
                check_against_oracle(ctx, oc, data, offsets=True, what=f"dense seed{seed}/{kw}")
            ctx.close()


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_multi_tile(seed):
    """Streams of 0.2-3 MB: lines straddle 16 KiB tile edges, several kernel passes, chunk mode."""
    rng = np.random.default_rng(2000 + seed)
    max_len = int(rng.choice([30, 150, 400, 5000, 40000]))
    nrec = int(rng.integers(50, 4000)) if max_len <= 400 else int(rng.integers(20, 120))
    dirty = float(rng.choice([0.0, 0.0, 0.001]))
    data = rand_stream(rng, n_records=nrec, max_len=max_len, dirty=dirty, crlf=bool(rng.random() < 0.2))
    for kw, pb in ((dict(), 0), (dict(check_ascii=True, check_quality=True), 64 * 1024),
                   (dict(check_ascii=True, check_quality=True), 16 * 1024)):
        ctx, oc = make_pair(batch_size=int(rng.choice([7, 4096])), pass_bytes=pb, emit_offsets=True, **kw)
        check_against_oracle(ctx, oc, data, offsets=True, what=f"mt seed{seed} {kw} pass={pb}")
        if EXPERIMENTS:   # the single-launch variants of an EXPERIMENTS build
            for mode, name in ((1, "look-back"), (2, "service"), (3, "two-level look-back")):
                ctx.set_option("single_pass", mode)
                check_against_oracle(ctx, oc, data, offsets=True, what=f"mt {name} seed{seed} {kw}")
            ctx.set_option("single_pass", 0)
        check_against_oracle(ctx, oc, data, is_eof=False, offsets=True, what=f"mt chunk-mode seed{seed}")
        ctx.set_option("force_dense", 1)
        check_against_oracle(ctx, oc, data, offsets=True, what=f"mt dense seed{seed}")
        ctx.close()


@pytest.mark.parametrize("single_pass", VARIANTS_LB)
def test_every_byte_value_is_classified_exactly(single_pass):
    """All 255 non-newline byte values at every alignment in id, sequence and quality lines: the newline
    detector (v_perm with the data as selector) and the strip logic must not mistake any of them."""
    allb = bytes(b for b in range(256) if b != 10)
    recs = []
    for i in range(300):
        rot = allb[i % 255:] + allb[:i % 255]
        body = rot[: 40 + (i * 7) % 215]
        recs.append(b"@" + rot[: (i * 3) % 60] + b"\n" + body + b"\n+\n" + body[::-1] + b"\n")
    data = b"".join(recs)
    ctx, ocfg = make_pair(single_pass=single_pass, check_ascii=False, check_quality=False)
    res, f = check_against_oracle(ctx, ocfg, data, what="all byte values")
    assert f.n_records == 300


def test_unterminated_last_record_that_fails_validation_is_not_consumed():
    """Found by tests/fuzz_campaign.py (seed 8002): the last record has no trailing newline (accepted, Q4) but its
    quality is out of range for the schema -- it must not be delivered and bytes_consumed stops before it."""
    data = b"@Tq_\nAGCGN\n+\n?hxj{\n@zbg\nNNAGAGGNACT\n+zbg\n8x+;@W4-)6z"
    for sp in VARIANTS_LB:
        ctx, ocfg = make_pair(batch_size=7, single_pass=sp, check_ascii=True, check_quality=True, quality_schema="solexa")
        res, f = check_against_oracle(ctx, ocfg, data, what="unterminated + invalid")
        assert f.n_records == 1 and f.term_code == 5 and f.consumed == 19 and int(res.bytes_consumed) == 19
        ctx.close()


def test_space_runs_across_tile_edges():
    """Header lines made of spaces that straddle tile boundaries: every id byte dropped exactly once."""
    rng = np.random.default_rng(7)
    parts = []
    for i in range(40):
        lead = b" " * int(rng.integers(0, 3000)); trail = b"\t" * int(rng.integers(0, 3000))
        rid = bytes(rng.integers(48, 123, int(rng.integers(0, 9))).astype(np.uint8))
        L = int(rng.integers(0, 5000))
        parts.append(b"@" + lead + rid + trail + b"\n" + b"A" * L + b"\n+\n" + b"I" * L + b"\n")
    data = b"".join(parts)
    for dense in (0, 1):
        for sp in VARIANTS:
            ctx, oc = make_pair(emit_offsets=True, check_ascii=True, check_quality=True, single_pass=sp)
            ctx.set_option("force_dense", dense)
            check_against_oracle(ctx, oc, data, offsets=True, what=f"space runs dense={dense} single_pass={sp}")
            ctx.close()


def test_tiny_records_take_serial_path_and_resize():
    """> 1020 newlines in a 16 KiB tile (records of 4-12 bytes): serial in-kernel path, and the
    per-record arrays are re-sized transparently."""
    data = b"@\n\n+\n\n" * 9000 + b"@a\nC\n+\n!\n" * 3000
    for sp in VARIANTS:
        ctx, oc = make_pair(emit_offsets=True, check_ascii=True, check_quality=True, single_pass=sp)
        res, f = check_against_oracle(ctx, oc, data, offsets=True, what=f"tiny single_pass={sp}")
        assert L_.lib().bzq_set_option(ctx.h, b"dense_tiles", 0) > 0  # dense tiles were used
        ctx.close()


def test_device_generator_matches_oracle():
    import torch
    ctx, _ = make_pair()
    for (nreads, L, lo, hi, sch) in ((64, 150, 33, 73, "generic"), (1000, 100, 0, 40, "sanger"), (10, 1, 5, 5, "solexa"), (3, 0, 0, 0, "generic")):
        nb = ctx.generate_synthetic_device(nreads, L, lo, hi, sch)
        t = torch.empty(nb + 16, dtype=torch.uint8, device="cuda")
        assert ctx.generate_synthetic_device(nreads, L, lo, hi, sch, t.data_ptr(), nb) == nb
        ref = O.generate_synthetic(nreads, L, L, lo, hi, sch)
        np.testing.assert_array_equal(t[:nb].cpu().numpy(), ref)
    # variable read lengths (config 4 style), whole file and record sub-ranges
    for (nreads, lo_len, hi_len, lo, hi, sch) in ((500, 5, 12, 33, 73, "generic"), (700, 200, 1300, 5, 30, "sanger"), (64, 0, 31, 0, 40, "sanger")):
        ref = O.generate_synthetic(nreads, lo_len, hi_len, lo, hi, sch)
        nb = ctx.generate_synthetic_device(nreads, lo_len, lo, hi, sch, max_len=hi_len)
        assert nb == ref.size
        t = torch.empty(nb + 16, dtype=torch.uint8, device="cuda")
        ctx.generate_synthetic_device(nreads, lo_len, lo, hi, sch, t.data_ptr(), nb, max_len=hi_len)
        np.testing.assert_array_equal(t[:nb].cpu().numpy(), ref)
        first, count = nreads // 3, nreads // 2
        before = ctx.generate_synthetic_device(nreads, lo_len, lo, hi, sch, count=first, max_len=hi_len)
        part = ctx.generate_synthetic_device(nreads, lo_len, lo, hi, sch, first=first, count=count, max_len=hi_len)
        ctx.generate_synthetic_device(nreads, lo_len, lo, hi, sch, t.data_ptr(), nb, first=first, count=count, max_len=hi_len)
        np.testing.assert_array_equal(t[:part].cpu().numpy(), ref[before:before + part])
    ctx.close()


@pytest.mark.parametrize("single_pass", VARIANTS)
@pytest.mark.parametrize("validate", [False, True])
def test_synthetic_150bp_medium(validate, single_pass):
    """Config 2/3 shape at a size the oracle parses in a second: 150k reads (47.7 MB)."""
    data = O.generate_synthetic(150_000, 150, 150, 33, 73, "generic")
    kw = dict(check_ascii=True, check_quality=True, quality_schema="sanger") if validate else {}
    ctx, oc = make_pair(emit_offsets=True, single_pass=single_pass, **kw)
    res, f = check_against_oracle(ctx, oc, data, offsets=True, what="150bp")
    assert int(res.n_records) == 150_000 and res.status == 6
    if validate:
        # negative variant: byte flips at record indices {0, 4095, 4096, R-1}
        for rec, pos, val, code in ((149_999, 30, 0x80, 4), (4096, 170, 0x1F, 5), (4095, 20, 0xC3, 4), (0, 200, 0x7F, 5)):
            bad = data.copy(); bad[rec * (data.size // 150_000) + pos] = val
            r2, f2 = check_against_oracle(ctx, oc, bad, what=f"flip rec {rec}")
            assert r2.status == code and int(r2.n_records) == rec
    ctx.close()


def test_long_reads_medium():
    """Config 4 shape: 200..19800 bp, high length variance (2000 reads, ~40 MB)."""
    data = O.generate_synthetic(2000, 200, 19800, 5, 30, "sanger")
    ctx, oc = make_pair(buffer_capacity=64 * 1024, check_ascii=True, check_quality=True, quality_schema="sanger", emit_offsets=True)
    check_against_oracle(ctx, oc, data, offsets=True, what="long reads")
    ctx.close()
    ctx, oc = make_pair(buffer_capacity=16 * 1024)   # the reference refuses records longer than its buffer
    res, f = check_against_oracle(ctx, oc, data, what="long reads small buffer")
    assert res.status == 8
    ctx.close()


def test_parser_api_mirrors_reference_tests(corpus_dir):
    """The reference's own API-level tests (tests/fastq/test_parser.mojo:122-215, tests/test_python_bindings.py:31-67)
    replayed on blazeseq_amd.FastqParser."""
    import blazeseq_amd as B
    content = b"@r1\nACGT\n+\n!!!!\n@r2\nTGCA\n+\n####\n@r3\nNNNN\n+\n!!!!\n"
    assert [len(b) for b in B.FastqParser(content, schema="generic", batch_size=2).batches()] == [2, 1]
    p = B.FastqParser(b"@a\nA\n+\n!\n@b\nB\n+\n!\n@c\nC\n+\n!\n@d\nD\n+\n!\n@e\nE\n+\n!\n", batch_size=2)
    assert [len(p.next_batch(2)) for _ in range(3)] == [2, 2, 1] and not p.has_more()
    b = B.FastqParser(b"@seq1\nACGT\n+\n!!!!\n", batch_size=4).next_batch(4)
    r = b.get_record(0)
    assert (len(b), r.id, r.sequence, r.quality) == (1, b"seq1", b"ACGT", b"!!!!")
    assert list(B.FastqParser(b"", batch_size=4).batches()) == []
    p = B.FastqParser(b"@r1\nA\n+\n!\n", batch_size=4)
    assert p.has_more(); p.next_batch(4); assert not p.has_more()
    data = open(os.path.join(corpus_dir, "example.fastq"), "rb").read()
    recs = list(B.FastqParser(data).records())
    assert [r.id for r in recs] == [b"EAS54_6_R1_2_1_413_324", b"EAS54_6_R1_2_1_540_792", b"EAS54_6_R1_2_1_443_348"]
    assert b"CCCTTCTTGTCTTCAGCGTTTCTCC" in recs[0].sequence
    p = B.FastqParser(os.path.join(corpus_dir, "example.fastq"))
    assert len(p.next_batch(2)) == 2 and len(p.next_batch(10)) == 1
    # FastqBatch layout, tests/fastq/test_record_batch.mojo:26-38
    b = B.FastqParser(b"@a\nAC\n+\n!!\n@b\nGT\n+\n!!\n").next_batch(4)
    assert b.num_records() == 2 and b.seq_len() == 4 and b._ends.tolist() == [2, 4]
    assert len(b._quality_bytes) == 4 and len(b._sequence_bytes) == 4 and b.quality_offset() == 33
    d = b.to_device()
    assert d.num_records == 2 and d.seq_len == 4 and d.total_id_bytes == 2
    back = d.copy_to_host()  # commented-out GPU round trip of tests/fastq/test_record_batch.mojo:144-378
    assert [(r.id, r.sequence, r.quality) for r in back.to_records()] == [(b"a", b"AC", b"!!"), (b"b", b"GT", b"!!")]
    # error text, tests/test_error_context.mojo:97-137
    p = B.FastqParser(b"@r1\nAT\n+\n!@\nr2\nGC\n+\n#$\n", config=B.ParserConfig(check_ascii=True, check_quality=True))
    assert len(p.next_record()) == 2
    with pytest.raises(B.ParseError, match="Record number: 2"):
        p.next_record()
    with pytest.raises(B.ParseError, match="Non ASCII letters found"):
        B.FastqParser(bytes([64, 114, 49, 10, 65, 200, 67, 10, 43, 10, 33, 33, 33, 10]), config=B.ParserConfig(check_ascii=True)).next_record()


def test_streaming_chunks_equal_one_shot():
    """Multi-chunk streaming (carry of the partial record/batch) yields the same batches as one chunk."""
    import blazeseq_amd as B
    data = O.generate_synthetic(30_000, 50, 150, 0, 40, "sanger")
    ref = [b for b in O.StreamParser(data, O.make_config(batch_size=1000)).batches()]
    for chunk in (1 << 16, 300_000, 1 << 30):
        # batches kept in a list outlive their chunk: they stay owned host objects like the reference's
        got = list(B.FastqParser(data, batch_size=1000, chunk_bytes=chunk).batches())
        assert [len(b) for b in got] == [len(b) for b in ref]
        for g, r in zip(got, ref):
            assert g._ends.tolist() == r.ends and g._id_ends.tolist() == r.id_ends
            assert g._sequence_bytes.tobytes() == r.seq_bytes and g._quality_bytes.tobytes() == r.qual_bytes
            assert g._id_bytes.tobytes() == r.id_bytes
        # a batch kept past the refill owns a host copy; to_device() uploads it (bzq_upload_batch), like the
        # reference's FastqBatch.to_device always does (record_batch.mojo:89-90, 404-411)
        for g, r in list(zip(got, ref))[:: max(1, len(ref) // 5)]:
            d = g.to_device()
            assert d.num_records == len(r) and d.seq_len == r.ends[-1] and d.total_id_bytes == r.id_ends[-1]
            back = d.copy_to_host()
            assert back._sequence_bytes.tobytes() == r.seq_bytes and back._quality_bytes.tobytes() == r.qual_bytes
            assert back._id_bytes.tobytes() == r.id_bytes and back._ends.tolist() == r.ends and back._id_ends.tolist() == r.id_ends
            assert int(d.histogram("sequence").sum()) == r.ends[-1]   # consumers run on an uploaded batch too
            d.release()
        # streaming use: to_device() inside the loop is a zero-copy view of the live chunk
        n = 0
        for g, r in zip(B.FastqParser(data, batch_size=1000, chunk_bytes=chunk).batches(), ref):
            d = g.to_device()
            assert d.num_records == len(r) and d.seq_len == r.ends[-1] and d.total_id_bytes == r.id_ends[-1]
            assert d.copy_to_host()._sequence_bytes.tobytes() == r.seq_bytes
            n += 1
        assert n == len(ref)


def _run_shards_on_one_gpu(data: np.ndarray, cuts, ocfg, single_pass=False, **kw):
    """Every shard goes through bzq_shard_scan / bzq_submit_shard on the one GPU; the halo exchange that
    RCCL does between ranks is a device-to-device copy here."""
    import torch
    from blazeseq_amd import sharded
    import blazeseq_amd as B
    bounds = [0, *cuts, data.size]
    P = len(bounds) - 1
    ctxs, bufs, sums = [], [], []
    for r in range(P):
        n = bounds[r + 1] - bounds[r]
        t = torch.zeros(n + (1 << 17), dtype=torch.uint8, device="cuda")
        t[:n] = torch.from_numpy(data[bounds[r]:bounds[r + 1]].copy()).cuda()
        torch.cuda.synchronize()   # torch's stream wrote it, the library's (non-blocking) stream reads it: seed 50820 of the campaign met the race
        ctx = B.Context(B.ParserConfig(**kw), "generic", 4096, 0)
        if single_pass:
            ctx.set_option("single_pass", int(single_pass))
        s = ctx.shard_scan(t.data_ptr(), n)
        ctxs.append(ctx); bufs.append(t)
        sums.append([int(s.n_bytes), int(s.n_newlines), *[int(x) for x in s.first_nl], int(s.first_byte), int(s.last_byte)])
        assert sums[-1][1] == int(np.count_nonzero(data[bounds[r]:bounds[r + 1]] == 10))
    plans = sharded.plan_shards(sums)
    total, ids, seqs, quals, ends = 0, [], [], [], []
    for r in range(P):
        n = bounds[r + 1] - bounds[r]
        p = plans[r]
        if p.halo_src >= 0:
            bufs[r][n:n + p.halo_bytes] = bufs[p.halo_src][:p.halo_bytes]
            torch.cuda.synchronize()
        is_last = all(s[0] == 0 for s in sums[r + 1:])
        ctxs[r].submit_shard(bufs[r].data_ptr(), n, p.halo_bytes, p.lines_before, p.prev_last_byte, bounds[r], is_last)
        res = ctxs[r].result()
        assert res.status in (0, 6), (r, res.status, ctxs[r].format_error())
        total += int(res.n_records)
        ids.append(res.id()); seqs.append(res.seq()); quals.append(res.qual())
        e = res.ends()
        ends.append(e + (ends[-1][-1] if ends and ends[-1].size else 0) if e.size else e)
    for c in ctxs:
        c.close()
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return total, cat(ids, np.uint8), cat(seqs, np.uint8), cat(quals, np.uint8), cat([e for e in ends if e.size], np.int64)


@pytest.mark.parametrize("single_pass", SHARD_VARIANTS)
@pytest.mark.parametrize("seed", range(6))
def test_shards_on_one_gpu(seed, single_pass):
    """bzq_shard_scan + k_head + bzq_submit_shard: byte-range shards cut anywhere reproduce the whole parse."""
    rng = np.random.default_rng(3000 + seed)
    max_len = int(rng.choice([40, 150, 3000]))
    data = np.frombuffer(rand_stream(rng, n_records=int(rng.integers(300, 1500)) if max_len < 1000 else 60,
                                     max_len=max_len, dirty=0.0, tail=0, crlf=bool(rng.random() < 0.2)), dtype=np.uint8)
    oc = O.make_config(check_ascii=True, check_quality=True)
    whole = O.flat_parse(data, oc)
    assert whole.term_code == O.EOF
    for P in (2, 3, 5):
        lo = 4 * (2 * max_len + 40)
        if data.size < P * lo * 2:
            continue
        cuts = sorted(int(x) for x in rng.integers(lo, data.size - lo, P - 1))
        cuts = [c for i, c in enumerate(cuts) if i == 0 or c - cuts[i - 1] > lo]
        total, ids, seqs, quals, ends = _run_shards_on_one_gpu(data, cuts, oc, single_pass=single_pass,
                                                              check_ascii=True, check_quality=True)
        assert total == whole.n_records, (P, cuts)
        np.testing.assert_array_equal(ids, whole.id_bytes)
        np.testing.assert_array_equal(seqs, whole.seq_bytes)
        np.testing.assert_array_equal(quals, whole.qual_bytes)
        np.testing.assert_array_equal(ends, whole.ends)


def test_shards_record_aligned_and_header_cut():
    """Cuts exactly at a record start, exactly after '@', and inside the '+' line."""
    rec = b"@read7 desc\nACGTACGT\n+read7\nIIIIIIII\n"
    data = np.frombuffer(rec * 800, dtype=np.uint8)
    oc = O.make_config()
    whole = O.flat_parse(data, oc)
    L = len(rec)
    for cut in (L * 400, L * 400 + 1, L * 400 + 12, L * 400 + 22, L * 400 + 27, L * 400 + L - 1):
        total, ids, seqs, quals, ends = _run_shards_on_one_gpu(data, [cut], oc)
        assert total == whole.n_records, cut
        np.testing.assert_array_equal(ids, whole.id_bytes)
        np.testing.assert_array_equal(quals, whole.qual_bytes)


def test_bitmap_only_pass_a_falls_back_exactly_when_an_id_is_stripped():
    """Pass A works from the newline bitmap alone under the hypothesis that no id loses bytes to _strip_spaces
    (utils.mojo:221-242); the emit measures every header line exactly and, when one contradicts the hypothesis, the host
    repeats the chunk with the exact pass A.  Clean ids (spaces INSIDE an id included) never fall back; a leading or a
    trailing space does, once; results are the oracle's either way, and identical with the option off."""
    from blazeseq_amd import _lib as L
    clean = b"".join(b"@read %d some description\n%s\n+\n%s\n" % (i, b"ACGT" * 30, b"IIII" * 30) for i in range(4000))
    lead = clean.replace(b"@read 2500 ", b"@ read 2500 ", 1)
    trail = clean.replace(b"@read 3999 some description\n", b"@read 3999 some description \t\n", 1)
    allsp = clean.replace(b"@read 17 some description\n", b"@   \n", 1)
    for data, want_fallbacks in ((clean, 0), (lead, 1), (trail, 1), (allsp, 1)):
        ctx, oc = make_pair(emit_offsets=True)
        check_against_oracle(ctx, oc, data, offsets=True, what="bitmap pass A")
        assert L.lib().bzq_set_option(ctx.h, b"stream_fallbacks", 0) == want_fallbacks
        check_against_oracle(ctx, oc, data, is_eof=False, offsets=True, what="bitmap pass A, chunk mode")
        ctx.set_option("pass_a_h", 0)
        before = L.lib().bzq_set_option(ctx.h, b"stream_fallbacks", 0)
        check_against_oracle(ctx, oc, data, offsets=True, what="exact pass A")
        assert L.lib().bzq_set_option(ctx.h, b"stream_fallbacks", 0) == before
        ctx.close()
