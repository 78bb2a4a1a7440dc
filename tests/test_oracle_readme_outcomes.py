"""Seven corpus files where the oracle does NOT give the outcome that the reference's own corpus README lists
(tests/test_data/fastq_parser/README.md:12,25,27,30,31,33,34, column "Current error").  Pinned here by name, with the source
lines that decide each of them, so that the difference is a stated fact and not an accident of the restatement:

* error_double_seq, error_trunc_at_plus, error_trunc_at_seq, error_trunc_in_seq, error_trunc_in_title -- README: "Quality and
  sequence line do not match in length"; oracle: "Separator line does not start with '+'" (SEP_NO_PLUS).
  `_validate_fastq_structure` (blazeseq/utils.mojo:448-462) tests '@' first, then '+', then the lengths, and in all five files
  the record that fails has a third line that does not start with '+' -- the length test is never reached.  The reference's
  corpus test accepts either message (tests/fastq/test_fastq_parser_correctness.mojo:21-58: `cor_len` OR `sep_line_start`
  OR "EOF"), so it cannot tell which one the binary gives; the source can.
* zero_length -- README: length mismatch; oracle: all 5 records parse (a zero-length read has zero-length sequence AND quality
  lines: utils.mojo:458-461 compares 0 with 0) and the stream ends with EOF, which the reference test accepts as "EOF"
  (test_fastq_parser_correctness.mojo:36,47; 742-747).
* example_dos -- README: "Parses successfully", which is what the oracle gives with the configuration the reference tests it
  with (validation off, test_fastq_parser_correctness.mojo:59-62, 149-154); WITH check_quality the '\\r' (13) that CRLF
  leaves at the end of every quality line is below every schema's lower bound (record.mojo:99-102) -> code 5 for record 1.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
CORPUS = os.path.join(HERE, "golden", "corpus")

SEP = b"Separator line does not start with '+'"
LEN = b"Quality and sequence line do not match in length"


def _parse(name, **kw):
    data = np.frombuffer(open(os.path.join(CORPUS, name), "rb").read(), dtype=np.uint8)
    return data, O.flat_parse(data, O.make_config(**kw), is_eof=True)


@pytest.mark.parametrize("name,records_before", [("error_double_seq.fastq", 3), ("error_trunc_at_plus.fastq", 4),
                                                 ("error_trunc_at_seq.fastq", 4), ("error_trunc_in_seq.fastq", 4),
                                                 ("error_trunc_in_title.fastq", 4)])
def test_plus_is_checked_before_the_lengths(name, records_before):
    data, f = _parse(name)
    assert f.term_code == 2 and f.n_records == records_before and f.term_msg.startswith(SEP)
    # the failing record's third line really does not start with '+', and its lengths differ as well: with the order of
    # utils.mojo:448-462 reversed the README's message would come out -- the order is what is pinned
    nl = np.flatnonzero(data == 10)
    start = 0 if records_before == 0 else int(nl[4 * records_before - 1]) + 1
    lines = bytes(data[start:]).split(b"\n")
    assert lines[0].startswith(b"@") and not lines[2].startswith(b"+")
    assert LEN not in f.term_msg
    # streaming restatement (BufferedReader + parser, line by line) agrees
    sp = O.StreamParser(data, O.make_config())
    with pytest.raises(O.OracleError) as e:
        while len(sp.next_batch(1)):
            pass
    assert str(e.value).encode("latin-1").startswith(SEP)


def test_zero_length_reads_are_records():
    data, f = _parse("zero_length.fastq")
    assert f.term_code == 6 and f.n_records == 5
    lens = np.diff(np.concatenate([[0], f.ends]))
    assert 0 in lens   # the zero-length read is delivered with empty sequence and quality


def test_example_dos_parses_without_validation_and_fails_quality_with_it():
    _, f = _parse("example_dos.fastq")
    assert f.term_code == 6 and f.n_records == 3
    assert bytes(f.seq_bytes).endswith(b"\r")   # CRLF: the '\r' stays in the fields, like the reference's spans
    _, v = _parse("example_dos.fastq", check_ascii=True, check_quality=True)
    assert v.term_code == 5 and v.term_record == 0 and b"Record number: 1" in v.term_msg


def test_golden_file_agrees():
    exp = json.load(open(os.path.join(HERE, "golden", "corpus_expected.json")))
    for name in ("error_double_seq.fastq", "error_trunc_at_plus.fastq", "error_trunc_at_seq.fastq", "error_trunc_in_seq.fastq",
                 "error_trunc_in_title.fastq"):
        assert exp[name]["default"]["term_code"] == 2
    assert exp["zero_length.fastq"]["default"]["term_code"] == 6
    assert exp["example_dos.fastq"]["default"]["term_code"] == 6 and exp["example_dos.fastq"]["validated_generic"]["term_code"] == 5
