"""The flat (newline-rank + window replay) restatement must agree with the streaming restatement
on every input, under adversarial buffer sizes.  The flat form is the spec the GPU kernels follow."""
import numpy as np
import pytest

from oracle import oracle as O
from fastq_fuzz import rand_stream


def compare(data, c):
    views, code, msg = O.StreamParser(data, c).stream_all()
    f = O.flat_parse(data, c, is_eof=True)
    assert f.n_records == len(views)
    assert f.term_code == code
    assert f.term_msg == msg
    S = Q = I = 0
    for r, v in enumerate(views):
        assert f.header_start[r] == v.rec_pos
        assert f.seq_start[r] == v.rec_pos + v.off[1]
        assert f.sep_start[r] == v.rec_pos + v.off[2]
        assert f.qual_start[r] == v.rec_pos + v.off[3]
        assert f.record_end[r] == v.rec_pos + v.off[4]
        assert f.id_start[r] == v.id_pos and f.id_len[r] == len(v.id)
        assert f.seq_bytes[S:S + len(v.seq)].tobytes() == v.seq
        assert f.qual_bytes[Q:Q + len(v.qual)].tobytes() == v.qual
        assert f.id_bytes[I:I + len(v.id)].tobytes() == v.id
        S += len(v.seq); Q += len(v.qual); I += len(v.id)
        assert f.ends[r] == Q and f.id_ends[r] == I


@pytest.mark.parametrize("seed", range(40))
def test_random_streams(seed):
    rng = np.random.default_rng(seed)
    data = rand_stream(rng, n_records=int(rng.integers(0, 40)), max_len=int(rng.integers(1, 60)),
                       dirty=float(rng.choice([0.0, 0.02, 0.1])), crlf=bool(rng.random() < 0.15))
    for cap in (16, 17, 31, 64, 100, 257, 4096, 256 * 1024):
        for growth, mx in ((False, 1 << 30), (True, 64), (True, 300), (True, 1 << 20)):
            for ca, cq, w in ((False, False, 0), (True, True, 0), (True, True, 32), (False, True, 16)):
                c = O.make_config(buffer_capacity=cap, buffer_growth_enabled=growth, buffer_max_capacity=mx,
                                  check_ascii=ca, check_quality=cq, simd_width=w,
                                  quality_schema="solexa" if seed % 5 == 0 else "generic")
                compare(data, c)


def test_window_quirks_explicit():
    # Q5: trailing junk while the window still starts at offset 0 -> BUFFER_EXCEEDED, not UNEXPECTED_EOF
    c = O.make_config()
    f = O.flat_parse(b"@only\nACGT", c)
    assert f.term_code == O.BUFFER_EXCEEDED and f.n_records == 0
    # ... but after at least one record (window base > 0) the zero-length read sets EOF first
    f = O.flat_parse(b"@a\nAC\n+\n!!\n@only\nACGT", c)
    assert f.term_code == O.UNEXPECTED_EOF and f.n_records == 1
    assert f.term_msg == b"Unexpected end of file in FASTQ record at phase 1"
    # Q4: last record without trailing newline accepted, structure check skipped
    f = O.flat_parse(b"@a\nAC\n+\n!!\n@b\nACGT\n+\n!!", c)
    assert f.term_code == O.EOF and f.n_records == 2 and f.ends.tolist() == [2, 4]
    assert f.seq_bytes.tobytes() == b"ACACGT"  # seq column keeps the true sequence bytes
    # blank remainder in QUAL phase -> `raise Error()` with an empty message
    f = O.flat_parse(b"@a\nAC\n+\n!!\n@b\nAC\n+\n \t", c)
    assert f.term_code == O.OTHER and f.term_msg == b"" and f.n_records == 1
    # trailing blank line after the last record
    f = O.flat_parse(b"@a\nAC\n+\n!!\n\n", c)
    assert f.term_code == O.UNEXPECTED_EOF and f.term_msg.endswith(b"phase 1")


def test_crlf_not_normalised():
    # Q6: seq/qual keep '\r', id loses it; with check_quality '\r' < LOWER -> code 5
    data = b"@id\r\nACGT\r\n+\r\n!!!!\r\n"
    f = O.flat_parse(data, O.make_config())
    assert f.n_records == 1 and f.seq_bytes.tobytes() == b"ACGT\r" and f.qual_bytes.tobytes() == b"!!!!\r"
    assert f.id_bytes.tobytes() == b"id"
    f = O.flat_parse(data, O.make_config(check_quality=True))
    assert f.term_code == O.QUALITY_OUT_OF_RANGE and f.n_records == 0


def test_simd_width_quirk():
    # Q9: '~' (126 == UPPER) inside the first floor(n/W)*W quality bytes is rejected by the SIMD body
    q = b"~" + b"I" * 31
    data = b"@r\n" + b"A" * 32 + b"\n+\n" + q + b"\n"
    assert O.flat_parse(data, O.make_config(check_quality=True, simd_width=0)).term_code == O.EOF
    assert O.flat_parse(data, O.make_config(check_quality=True, simd_width=32)).term_code == O.QUALITY_OUT_OF_RANGE
    assert O.flat_parse(data, O.make_config(check_quality=True, simd_width=64)).term_code == O.EOF


def test_chunk_mode_leaves_tail():
    data = b"@a\nAC\n+\n!!\n@b\nACGT\n+\n!!"
    f = O.flat_parse(data, O.make_config(), is_eof=False)
    assert f.n_records == 1 and f.consumed == 11 and f.term_code == O.OK
