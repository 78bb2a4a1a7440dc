"""-m gpu: the reference's Python-binding tests (tests/test_python_bindings.py:31-120) replayed on
blazeseq_amd.parser() -- same file (example.fastq from the reference's corpus), same expectations."""
import os

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FASTQ_PATH = os.path.join(HERE, "golden", "corpus", "example.fastq")


def test_create_parser_and_next_record():
    import blazeseq_amd as blazeseq
    parser = blazeseq.parser(FASTQ_PATH, "generic")
    assert parser.has_more()
    count = 0
    while True:
        try:
            rec = parser.next_record()
            count += 1
            if count == 1:
                assert rec.id == "EAS54_6_R1_2_1_413_324"
                assert "CCCTTCTTGTCTTCAGCGTTTCTCC" in rec.sequence
                seq_len = len(rec.sequence)
                assert seq_len > 0
                assert len(rec) >= seq_len
                assert len(rec.phred_scores) >= seq_len
                assert rec.phred_scores[0] == ord(rec.quality[0]) - 33
        except Exception as e:
            if "EOF" in str(e):
                break
            raise
    assert count == 3


def test_next_batch_and_get_record():
    import blazeseq_amd as blazeseq
    parser = blazeseq.parser(FASTQ_PATH, "generic")
    batch = parser.next_batch(2)
    assert batch.num_records() == 2
    assert batch.get_record(0).id == "EAS54_6_R1_2_1_413_324"
    assert batch.get_record(1).id == "EAS54_6_R1_2_1_540_792"
    batch2 = parser.next_batch(10)
    assert batch2.num_records() == 1
    assert batch2.get_record(0).id == "EAS54_6_R1_2_1_443_348"


def test_eof_raises():
    import blazeseq_amd as blazeseq
    parser = blazeseq.parser(FASTQ_PATH, "generic")
    for _ in range(3):
        parser.next_record()
    with pytest.raises(Exception, match="EOF"):
        parser.next_record()


def test_parser_and_batch_iterator_protocols():
    import blazeseq_amd as blazeseq
    recs = list(blazeseq.parser(FASTQ_PATH, "generic"))
    assert len(recs) == 3 and recs[0].id == "EAS54_6_R1_2_1_413_324"
    batch = blazeseq.parser(FASTQ_PATH, "generic").next_batch(2)
    got = [r.id for r in batch]
    assert got == ["EAS54_6_R1_2_1_413_324", "EAS54_6_R1_2_1_540_792"]
    p = blazeseq.create_parser(FASTQ_PATH)
    assert [b.num_records() for b in p.batches_with_size(2)] == [2, 1]
    assert sum(b.num_records() for b in blazeseq.parser(FASTQ_PATH).batches) == 3
    assert [r.id for r in blazeseq.parser(FASTQ_PATH).records][2] == "EAS54_6_R1_2_1_443_348"


def test_records_before_a_failing_record_are_delivered_then_the_error(tmp_path):
    """Record-wise iteration pulls records in bulk; a failing record must still surface exactly after its
    predecessors (tests/test_error_context.mojo:97-137 semantics)."""
    import blazeseq_amd as blazeseq
    recs = [b"@r%d\nACGT\n+\nIIII\n" % i for i in range(6000)]
    recs[5000] = b"r5000\nACGT\n+\nIIII\n"
    path = tmp_path / "bad.fastq"
    path.write_bytes(b"".join(recs))
    p = blazeseq.parser(str(path))
    n = 0
    with pytest.raises(blazeseq.ParseError, match="Record number: 5001") as err:
        for _ in range(7000):
            p.next_record()
            n += 1
    assert n == 5000 and "does not start with '@'" in str(err.value)
    # mixing next_record and next_batch keeps the order
    q = blazeseq.parser(str(path))
    assert q.next_record().id == "r0"
    b = q.next_batch(3)
    assert [r.id for r in b] == ["r1", "r2", "r3"] and q.next_record().id == "r4"
