"""Host side of the FASTA mirror without a GPU: ``FastaRecord`` / ``Definition`` (blazeseq_amd/fasta.py) against the
reference's record tests (tests/fasta/test_fasta_parser.mojo:835-1028, fasta/record.mojo:11-144).  Where the reference
test parses first, the records come from the oracle's restatement of the parser (the GPU parser replays the same tests
in tests/test_gpu_fasta.py)."""
from oracle import fasta as F
from blazeseq_amd.fasta import FastaRecord, Definition, FastaParserConfig


def _records(data: bytes):
    recs, code, _ = F.StreamFastaParser(data).all_records()
    assert code == F.EOF
    return [FastaRecord(i, s) for i, s in recs]


def test_byte_len_len_and_accessors():   # :835, :849, :895, :906
    assert _records(b">abc\nACGT\n")[0].byte_len() == 10
    assert len(_records(b">id1\nACGT\nACGT\n")[0]) == 8
    assert _records(b">id\nAC\nGT\n")[0].byte_len() == 9
    r = _records(b">myid\nGATTACA\n")[0]
    assert (r.id, r.sequence, len(r), r.byte_len()) == (b"myid", b"GATTACA", 7, 14)


def test_write_format_and_wrapping():   # :859, record.mojo:107-124
    assert _records(b">id1\nACGT\n")[0].write() == b">id1\nACGT\n" == repr(_records(b">id1\nACGT\n")[0]).encode()
    r = FastaRecord("x", "ACGT" * 40)
    lines = r.write().split(b"\n")
    assert lines[0] == b">x" and [len(l) for l in lines[1:-1]] == [60, 60, 40] and lines[-1] == b""
    assert r.write(0) == b">x\n" + b"ACGT" * 40 + b"\n"          # width <= 0: one line
    assert FastaRecord("e", "").write() == b">e\n"


def test_equality_is_on_the_sequence_only():   # :873, :886, record.mojo:133-142
    assert FastaRecord("id1", "ACGT") == FastaRecord("id2", "ACGT")
    assert FastaRecord("id1", "ACGT") != FastaRecord("id1", "TTAA")
    assert hash(FastaRecord("a", "ACGT")) == hash(FastaRecord("b", "ACGT"))
    assert len({FastaRecord("a", "ACGT"), FastaRecord("b", "ACGT"), FastaRecord("c", "AC")}) == 2


def test_definition_splits_id_and_description():   # record.mojo:86-99, definition.mojo
    assert FastaRecord("id1", "A").definition() == Definition(b"id1", None)
    assert FastaRecord("id1 description here", "A").definition() == Definition(b"id1", b"descriptionhere")
    assert FastaRecord("gi|1|x  two  gaps ", "A").definition() == Definition(b"gi|1|x", b"twogaps")


def test_roundtrip_read_write_read():   # :957-1028
    for data in (b">id1\nACGT\n", b">id1\nACGT\n>id2\nTTAA\n>id3\nGGCC\n", b">id1 description here\nACGT\n", b">seq1\nACG\nTTA\nGG\n",
                 b">long\n" + b"ACGT" * 100 + b"\n"):
        original = _records(data)
        again = _records(b"".join(r.write() for r in original))
        assert [(r.id, r.sequence) for r in original] == [(r.id, r.sequence) for r in again]
    assert _records(b">seq1\nACG\nTTA\nGG\n")[0].sequence == b"ACGTTAGG"
    assert len(_records(b">long\n" + b"ACGT" * 100 + b"\n")[0].sequence) == 400


def test_config_defaults():   # fasta/parser.mojo:24-35, CONSTS.mojo:26
    c = FastaParserConfig()
    assert c.check_ascii is False and c.line_capacity == 256 * 1024
