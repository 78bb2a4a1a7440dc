"""Oracle vs committed golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py) and
vs the independent kseq record/base counts (oracle/_ref/kseq_runner is built
from the reference's own benchmark/fastq-parser/kseq_runner/main.c)."""
import hashlib
import json
import os

import pytest

from oracle import oracle as O
import importlib.util

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)

CORPUS = json.load(open(os.path.join(HERE, "golden", "corpus_expected.json")))
SYN = json.load(open(os.path.join(HERE, "golden", "synthetic_expected.json")))

# Independent counts, obtained once with kseq_runner; kseq strips '\r', so
# example_dos reports 75 where BlazeSeq semantics give 78 (Q6).
KSEQ = {"example.fastq": (3, 75), "illumina_example.fastq": (250, 9000), "test1_sanger.fastq": (250, 65558),
        "test2_solexa.fastq": (5, 125), "test3_illumina.fastq": (25, 625), "longreads_as_sanger.fastq": (10, 3665),
        "misc_dna_as_sanger.fastq": (4, 153), "misc_rna_as_sanger.fastq": (4, 153), "sanger_93.fastq": (1, 94),
        "sanger_faked.fastq": (1, 41), "solexa_faked.fastq": (1, 46), "illumina_faked.fastq": (1, 41),
        "illumina_full_range_as_sanger.fastq": (2, 126), "sanger_full_range_as_sanger.fastq": (2, 188),
        "solexa_full_range_as_sanger.fastq": (2, 136), "wrapping_as_sanger.fastq": (3, 410),
        "solexa_example.fastq": (5, 125)}


@pytest.mark.parametrize("name", sorted(CORPUS))
def test_corpus_file(name, corpus_dir):
    data = open(os.path.join(corpus_dir, name), "rb").read()
    e = CORPUS[name]
    assert len(data) == e["size"]
    sc = e["schema"]
    cfgs = {"default": O.make_config(),
            "validated_generic": O.make_config(check_ascii=True, check_quality=True),
            "validated_schema": O.make_config(check_ascii=True, check_quality=True, quality_schema=sc),
            "validated_schema_simd32": O.make_config(check_ascii=True, check_quality=True, quality_schema=sc, simd_width=32),
            "cap64": O.make_config(buffer_capacity=64),
            "cap64_growth": O.make_config(buffer_capacity=64, buffer_growth_enabled=True, buffer_max_capacity=1 << 20)}
    for key, cfg in cfgs.items():
        assert MG.entry(data, cfg) == e[key], key
        views, code, msg = O.StreamParser(data, cfg).stream_all()
        assert (len(views), code, msg.decode("latin-1")) == (e[key]["n_records"], e[key]["term_code"], e[key]["term_msg"]), key
    if name in KSEQ:
        assert (e["default"]["n_records"], e["default"]["bases"]) == KSEQ[name]
        assert tuple(e.get("kseq", KSEQ[name])) == KSEQ[name]


def test_valid_files_parse_clean_with_their_schema():
    # tests/fastq/test_fastq_parser_correctness.mojo:142-444 run these with validation off; the README
    # marks them "Parses successfully".  With validation ON and the file's own schema they are clean too
    # (scalar-inclusive quality bounds), except the CRLF file (Q6).
    for name, e in CORPUS.items():
        if name.startswith(("error_", "empty", "zero_length", "tricky")) or "invalid" in name or "original_sanger" in name and ("longreads" in name or "wrapping" in name):
            continue
        assert e["default"]["term_code"] == O.EOF, name
        if name != "example_dos.fastq" and ("_as_" not in name or name.endswith("as_" + {"sanger": "sanger", "solexa": "solexa", "illumina_1.3": "illumina"}.get(e["schema"], "x") + ".fastq")):
            assert e["validated_schema"]["term_code"] == O.EOF, name


@pytest.mark.parametrize("key", sorted(SYN))
def test_synthetic_generator_pinned(key):
    e = SYN[key]
    buf = O.generate_synthetic(*e["args"])
    assert buf.size == e["size"]
    assert hashlib.sha256(buf.tobytes()).hexdigest() == e["sha256"]
    assert buf[:80].tobytes().decode("latin-1") == e["first_bytes"]


def test_synthetic_generator_by_hand():
    # utils.mojo:736-828 evaluated by hand for record 0 of (2 reads, len 3, phred 10..10, generic):
    # header "@read_0", LUT "GCGCATAT", s0 = 1442695040888963407 & (2^63-1); quality: q_range 0 ->
    # noise_amp 1, mean 10, phred clamped to [10,10] -> byte 43 '+'
    buf = O.generate_synthetic(2, 3, 3, 10, 10, "generic").tobytes()
    M = (1 << 63) - 1
    s = (0 * 6364136223846793005 + 1442695040888963407) & M
    seq = b""
    for _ in range(3):
        s = (s * 6364136223846793005 + 1442695040888963407) & M
        seq += b"GCGCATAT"[(s >> 33) % 8:(s >> 33) % 8 + 1]
    assert buf.startswith(b"@read_0\n" + seq + b"\n+\n+++\n@read_1\n")
    # compute_num_reads_for_size, utils.mojo:640-678: 3 GiB of 100 bp reads -> 14.7 M
    n = O.compute_num_reads_for_size(3 * 1024 ** 3, 100, 100)
    assert n == 3 * 1024 ** 3 // (6 + 8 + 1 + 204)


def _spec_record(i, nd, min_len, max_len, min_phred, max_phred, schema):
    """Record i of SURVEY.md Appendix B (utils.mojo:707-917) written out a second time, in plain Python, independent of oracle/:
    the byte-exact definition every BASELINE input flows from (nd = digits of the header number)."""
    M = (1 << 63) - 1
    lower, upper, offset = {"generic": (33, 126, 33), "sanger": (33, 126, 33), "solexa": (59, 126, 64), "illumina_1.3": (64, 126, 64),
                            "illumina_1.5": (66, 126, 64), "illumina_1.8": (33, 126, 33)}[schema]
    lut = b"GCGCATAT"      # gc_bias 0.5 -> 4 of the 8 slots alternate G, C, the rest A, T
    q_range = max_phred - min_phred
    noise_amp = q_range // 6 + 1
    out = bytearray()
    L = min_len if min_len == max_len else min_len + ((i * 31 + 7) % (max_len - min_len + 1))
    out += b"@read_" + str(i).zfill(nd).encode() + b"\n"
    st = (i * 6364136223846793005 + 1442695040888963407) & M
    for _ in range(L):
        st = (st * 6364136223846793005 + 1442695040888963407) & M
        out.append(lut[(st >> 33) % 8])
    out += b"\n+\n"
    r = (i * 2654435761 + 1013904223) & M
    for p in range(L):
        mean = max_phred if L == 1 else max_phred - (q_range * p + (L - 1) // 2) // (L - 1)
        r = (r * 1664525 + 1013904223) & M
        phred = mean + ((r >> 17) % (2 * noise_amp + 1)) - noise_amp
        phred = min(max(phred, min_phred), max_phred)
        out.append(min(max(offset + phred, lower), upper))
    out += b"\n"
    return bytes(out)


def _generator_spec(num_reads, min_len, max_len, min_phred, max_phred, schema):
    """the pin has to look at more than record 0 of a 3-base read -- variable lengths, the quality ramp,
    noise_amp, the clamp to the schema, 7- and 9-digit headers."""
    nd = 1 if num_reads <= 1 else len(str(num_reads - 1))
    return b"".join(_spec_record(i, nd, min_len, max_len, min_phred, max_phred, schema) for i in range(num_reads))


@pytest.mark.parametrize("args", [
    (1200, 150, 150, 33, 73, "generic"),          # the bench's records: 4-digit headers here
    (3, 150, 150, 33, 73, "generic"),
    (700, 5, 12, 0, 40, "sanger"),                # variable lengths (test_parser.mojo:228-288's shape), ramp + noise + clamp
    (300, 200, 19_800, 5, 30, "sanger"),          # config 4's long reads
    (40, 1, 1, 10, 10, "illumina_1.3"),           # L = 1: the ramp's special case; another offset
    (25, 30, 60, 0, 62, "solexa"),                # clamp to the schema's LOWER
])
def test_synthetic_generator_against_an_independent_restatement(args):
    assert O.generate_synthetic(*args).tobytes() == _generator_spec(*args)


def test_synthetic_generator_header_width_follows_the_read_count():
    """7-digit headers for the 10 M-read input (config 2), 9 digits for the 625 M-read one (config 5): record i depends only on i
    and on the digit count, so single records of those files are checked without generating the files."""
    for total, digits in ((10_000_000, 7), (625_000_000, 9)):
        for i in (0, 4095, 4096, total - 1):
            one = O.generate_synthetic(total, 150, 150, 33, 73, "generic", first=i, count=1).tobytes()
            assert one == _spec_record(i, digits, 150, 150, 33, 73, "generic")
