"""Pins the CPU oracle against the reference's own known-answer tests.

Each test transcribes the literal expectation of one reference test (file:line cited; paths are
relative to /root/reference).  The reference cannot run here (Mojo), so these literals are what
anchors the oracle; the GPU path is then checked against the oracle.
"""
import pytest

from oracle import oracle as O

R2 = b"@r1\nACGT\n+\n!!!!\n@r2\nTGCA\n+\n####\n"


def cfg(**kw):
    return O.make_config(**kw)


def non_ascii_fastq():
    # tests/fastq/test_parser.mojo:23-39
    return bytes([ord("@"), ord("r"), ord("1"), 10, ord("A"), 200, ord("C"), 10, ord("+"), 10,
                  ord("!"), ord("!"), ord("!"), 10])


def test_record_parser_for_loop():
    # tests/fastq/test_parser.mojo:42-67
    views = list(O.StreamParser(R2).views())
    assert len(views) == 2
    assert views[0].id == b"r1" and views[0].seq == b"ACGT"
    assert views[1].id == b"r2"


def test_stop_iteration_then_empty():
    # tests/fastq/test_parser.mojo:70-84
    p = O.StreamParser(b"@r1\nACGT\n+\n!!!!\n")
    assert len(list(p.views())) == 1
    assert len(list(p.views())) == 0


def test_ascii_validation_enabled_and_disabled():
    # tests/fastq/test_parser.mojo:87-114, 561-571
    with pytest.raises(O.OracleError, match="Non ASCII letters found"):
        O.StreamParser(non_ascii_fastq(), cfg(check_ascii=True)).next_view()
    v = O.StreamParser(non_ascii_fastq(), cfg(check_ascii=False)).next_view()
    assert v.id == b"r1"


def test_batched_parser_for_loop():
    # tests/fastq/test_parser.mojo:122-140 : batch_size=2, 3 records -> 2 + 1
    content = b"@r1\nACGT\n+\n!!!!\n@r2\nTGCA\n+\n####\n@r3\nNNNN\n+\n!!!!\n"
    batches = list(O.StreamParser(content, cfg(batch_size=2)).batches())
    assert [len(b) for b in batches] == [2, 1]


def test_batch_size_respected_and_has_more():
    # tests/fastq/test_parser.mojo:143-164
    content = b"@a\nA\n+\n!\n@b\nB\n+\n!\n@c\nC\n+\n!\n@d\nD\n+\n!\n@e\nE\n+\n!\n"
    p = O.StreamParser(content, cfg(batch_size=2))
    assert [len(p.next_batch(2)) for _ in range(3)] == [2, 2, 1]
    assert not p.has_more()


def test_single_batch_content():
    # tests/fastq/test_parser.mojo:167-182
    b = O.StreamParser(b"@seq1\nACGT\n+\n!!!!\n", cfg(batch_size=4)).next_batch(4)
    assert len(b) == 1
    assert b.get_record(0) == (b"seq1", b"ACGT", b"!!!!")


def test_empty_input_yields_no_batches():
    # tests/fastq/test_parser.mojo:185-197
    assert list(O.StreamParser(b"", cfg(batch_size=4)).batches()) == []


def test_has_more_before_and_after():
    # tests/fastq/test_parser.mojo:200-215
    p = O.StreamParser(b"@r1\nA\n+\n!\n", cfg(batch_size=4))
    assert p.has_more()
    p.next_batch(4)
    assert not p.has_more()


def test_generate_synthetic_counts_and_lengths():
    # tests/fastq/test_parser.mojo:228-255
    buf = O.generate_synthetic(20, 5, 12, 2, 25, "generic")
    assert buf.size > 0
    total = 0
    for b in O.StreamParser(buf, cfg(batch_size=8)).batches():
        total += len(b)
        for i in range(len(b)):
            assert 5 <= len(b.get_record(i)[1]) <= 12
    assert total == 20


def test_generator_piecewise_equals_whole():
    # tests/fastq/test_parser.mojo:258-288 (writer path == buffer path): here record-range
    # generation must concatenate to the whole-buffer output byte for byte.
    whole = O.generate_synthetic(12, 5, 11, 2, 40, "generic")
    parts = [O.generate_synthetic(12, 5, 11, 2, 40, "generic", first=i, count=1) for i in range(12)]
    assert b"".join(p.tobytes() for p in parts) == whole.tobytes()


def test_fast_path_all_lines_in_buffer():
    # tests/fastq/test_parser.mojo:294-311 : buffer_capacity=256
    p = O.StreamParser(b"@r1\nACGT\n+\n!!!!\n@r2\nTGCA\n+\n!!!!\n", cfg(buffer_capacity=256))
    r1, r2 = p.next_view(), p.next_view()
    assert (r1.id, r1.seq, r1.qual) == (b"r1", b"ACGT", b"!!!!")
    assert (r2.id, r2.seq) == (b"r2", b"TGCA")
    with pytest.raises(O.OracleError, match="EOF"):
        p.next_view()


def test_record_spans_chunks_small_buffer():
    # tests/fastq/test_parser.mojo:329-341, 349-403, 492-511 : buffer_capacity=32
    p = O.StreamParser(b"@r1\nACGT\n+\n!!!!\n", cfg(buffer_capacity=32))
    r = p.next_view()
    assert (r.id, r.seq, r.qual) == (b"r1", b"ACGT", b"!!!!")
    with pytest.raises(O.OracleError, match="EOF"):
        p.next_view()
    p = O.StreamParser(b"@r1\nA\n+\n!\n@r2\nB\n+\n!\n@r3\nC\n+\n!\n", cfg(buffer_capacity=32))
    got = [(v.id, v.seq, v.qual) for v in (p.next_view(), p.next_view(), p.next_view())]
    assert got == [(b"r1", b"A", b"!"), (b"r2", b"B", b"!"), (b"r3", b"C", b"!")]
    with pytest.raises(O.OracleError, match="EOF"):
        p.next_view()
    got = [(v.id, v.seq, v.qual) for v in O.StreamParser(b"@a\nAC\n+\n!!\n@b\nTG\n+\n##\n", cfg(buffer_capacity=32)).views()]
    assert got == [(b"a", b"AC", b"!!"), (b"b", b"TG", b"##")]


def test_empty_input_next_view_eof():
    # tests/fastq/test_parser.mojo:459-467
    with pytest.raises(O.OracleError, match="EOF"):
        O.StreamParser(b"", cfg(buffer_capacity=256)).next_view()


def test_invalid_header_and_length_mismatch():
    # tests/fastq/test_parser.mojo:470-489
    with pytest.raises(O.OracleError, match="Sequence id line does not start with '@'"):
        O.StreamParser(b"r1\nACGT\n+\n!!!!\n", cfg(buffer_capacity=256)).next_view()
    with pytest.raises(O.OracleError, match="Quality and sequence line do not match in length"):
        O.StreamParser(b"@r1\nACGT\n+\n!!!\n", cfg(buffer_capacity=256)).next_view()


LONG = b"@id\n" + b"A" * 20 + b"\n+\n" + b"!" * 20 + b"\n"


def test_long_line_with_growth():
    # tests/fastq/test_parser.mojo:527-545 : capacity 16, growth on, max 256
    p = O.StreamParser(LONG, cfg(buffer_capacity=16, buffer_growth_enabled=True, buffer_max_capacity=256))
    r = p.next_view()
    assert r.id == b"id" and r.seq == b"A" * 20 and r.qual == b"!" * 20
    with pytest.raises(O.OracleError, match="EOF"):
        p.next_view()


def test_long_line_without_growth():
    # tests/fastq/test_parser.mojo:548-558
    with pytest.raises(O.OracleError, match="record exceeds buffer capacity"):
        O.StreamParser(LONG, cfg(buffer_capacity=16, buffer_growth_enabled=False)).next_view()


def test_fastq_batch_layout():
    # tests/fastq/test_record_batch.mojo:26-38 : ("AC","!!"),("GT","!!") -> _ends == [2,4]
    b = O.StreamParser(b"@a\nAC\n+\n!!\n@b\nGT\n+\n!!\n").next_batch(4)
    assert b.n == 2 and b.ends == [2, 4] and b.ends[-1] == 4
    assert len(b.qual_bytes) == 4 and len(b.seq_bytes) == 4
    assert b.quality_offset == 33
    # tests/fastq/test_record_batch.mojo:60-141 : get_record round trips
    assert b.get_record(0) == (b"a", b"AC", b"!!") and b.get_record(1) == (b"b", b"GT", b"!!")


def test_validator_ascii_and_quality():
    # tests/fastq/test_fastq_record.mojo:126-181 : byte 128 in id -> ascii error;
    # quality "   " (32) invalid, "!!!" valid for generic 33..126
    bad_id = b"@r" + bytes([128]) + b"\nACG\n+\n!!!\n"
    with pytest.raises(O.OracleError, match="Non ASCII letters found"):
        O.StreamParser(bad_id, cfg(check_ascii=True)).next_view()
    with pytest.raises(O.OracleError, match="Corrupt quality score according to provided schema"):
        O.StreamParser(b"@r\nACG\n+\n   \n", cfg(check_quality=True)).next_view()
    assert O.StreamParser(b"@r\nACG\n+\n!!!\n", cfg(check_quality=True)).next_view().qual == b"!!!"


def test_error_context_record_and_line_numbers():
    # tests/test_error_context.mojo:57-149
    invalid_id = b"r1\nATCG\n+\n!@#$\n"
    with pytest.raises(O.OracleError) as e:
        O.StreamParser(invalid_id, cfg(check_ascii=True, check_quality=True)).next_view()
    assert "Record number" in str(e.value) and "Line number" in str(e.value)
    mismatch = b"@r1\nATCG\n+\n!@#\n"
    with pytest.raises(O.OracleError, match="Record number"):
        O.StreamParser(mismatch, cfg(check_ascii=True, check_quality=True)).next_view()
    assert len(list(O.StreamParser(invalid_id, cfg(check_ascii=True, check_quality=True)).views())) == 0
    two = b"@r1\nAT\n+\n!@\nr2\nGC\n+\n#$\n"
    p = O.StreamParser(two, cfg(check_ascii=True, check_quality=True))
    assert len(p.next_view().seq) == 2
    with pytest.raises(O.OracleError, match="Record number: 2"):
        p.next_view()


def test_python_bindings_example_fastq(corpus_dir):
    # tests/test_python_bindings.py:31-67
    data = open(corpus_dir + "/example.fastq", "rb").read()
    views = list(O.StreamParser(data).views())
    assert [v.id for v in views] == [b"EAS54_6_R1_2_1_413_324", b"EAS54_6_R1_2_1_540_792", b"EAS54_6_R1_2_1_443_348"]
    assert b"CCCTTCTTGTCTTCAGCGTTTCTCC" in views[0].seq
    p = O.StreamParser(data)
    assert len(p.next_batch(2)) == 2
    assert len(p.next_batch(10)) == 1


def test_schema_table():
    # fastq/quality_schema.mojo:26-31, utils.mojo:612-637
    assert O.schema("generic") == (33, 126, 33, True)
    assert O.schema("sanger") == (33, 126, 33, True)
    assert O.schema("solexa") == (59, 126, 64, True)
    assert O.schema("illumina_1.3") == (64, 126, 64, True)
    assert O.schema("illumina_1.5") == (66, 126, 64, True)
    assert O.schema("illumina_1.8") == (33, 126, 33, True)
    assert O.schema("nonsense") == (33, 126, 33, False)


def test_error_message_exact_text():
    # errors.mojo:178-192 + parser.mojo:332-338: first record -> record 1, line 1, position 0 omitted
    with pytest.raises(O.OracleError) as e:
        O.StreamParser(b"r1\nACGT\n+\n!!!!\n").next_view()
    assert e.value.message == (b"Sequence id line does not start with '@'\n  Record number: 1\n  Line number: 1"
                               b"\n  Record snippet: r1\nACGT\n+\n!!!!\n")
    with pytest.raises(O.OracleError) as e:
        p = O.StreamParser(b"@a\nAC\n+\n!!\nr2\nGC\n+\n#$\n")
        p.next_view(); p.next_view()
    assert e.value.message == (b"Sequence id line does not start with '@'\n  Record number: 2\n  Line number: 5"
                               b"\n  File position: 11\n  Record snippet: r2\nGC\n+\n#$\n")
    # validation error: errors.mojo:223-234, parser.mojo:162-169 (no line number, no position)
    with pytest.raises(O.OracleError) as e:
        O.StreamParser(b"@r\nACG\n+\n   \n", cfg(check_quality=True)).next_view()
    assert e.value.message == (b"Corrupt quality score according to provided schema\n  Record number: 1"
                               b"\n  Record snippet: r\nACG")
