"""The drop-in boundary used from plain C (tests/c_driver/bzq_cat.c): the header compiles as C11 with -Werror
(CPU test) and the resulting program, with no Python or torch in the process, reproduces the oracle's records,
batch count, terminal status and error text (GPU test)."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DRV = os.path.join(HERE, "c_driver")
LIB = os.path.join(HERE, "..", "blazeseq_amd", "libblazeseq_hip.so")


def _build():
    if not os.path.exists(LIB):
        pytest.skip("libblazeseq_hip.so not built")
    subprocess.run(["make", "-C", DRV], check=True, capture_output=True)
    return os.path.join(DRV, "bzq_cat")


def test_header_is_plain_c_and_the_driver_links():
    exe = _build()
    assert os.access(exe, os.X_OK)
    # the header alone, strictest mode, as C and as C++
    for comp, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        src = '#include "blazeseq_hip.h"\nint main(void) { return (int)sizeof(bzq_chunk) == 0; }\n'
        r = subprocess.run([comp, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only",
                            "-I", os.path.join(HERE, "..", "include"), "-x", "c" if comp == "gcc" else "c++", "-"],
                           input=src.encode(), capture_output=True)
        assert r.returncode == 0, r.stderr.decode()


def _expect(data: bytes, batch: int, check: bool):
    from oracle import oracle as O
    cfg = O.make_config(batch_size=batch, check_ascii=check, check_quality=check)
    f = O.flat_parse(np.frombuffer(data, dtype=np.uint8), cfg, is_eof=True)
    lines, e0, i0 = [], 0, 0
    for r in range(f.n_records):
        e1, i1 = int(f.ends[r]), int(f.id_ends[r])
        lines.append(f.id_bytes[i0:i1].tobytes() + b"\t" + f.seq_bytes[e0:e1].tobytes() + b"\t" + f.qual_bytes[e0:e1].tobytes() + b"\n")
        e0, i0 = e1, i1
    nb = (f.n_records + batch - 1) // batch
    out = b"".join(lines) + b"# records=%d batches=%d status=%d\n" % (f.n_records, nb, f.term_code)
    if f.term_code != 6:
        out += b"# error: " + f.term_msg + b"\n"
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["example", "clean_multi_chunk", "error_late", "truncated", "validated_bad_quality"])
def test_c_driver_reproduces_the_oracle(case, tmp_path):
    from fastq_fuzz import rand_stream
    exe = _build()
    rng = np.random.default_rng(7)
    batch, chunk, check = 4096, 0, 0
    if case == "example":
        data = open(os.path.join(HERE, "golden", "corpus", "example.fastq"), "rb").read()
        batch = 2
    elif case == "clean_multi_chunk":
        data = rand_stream(rng, n_records=3000, max_len=100, dirty=0.0, tail=0)
        batch, chunk = 100, 8192
    elif case == "error_late":
        recs = [b"@r%d\nACGTAC\n+\nIIIIII\n" % i for i in range(4000)]
        recs[3210] = b"@r3210\nACGTAC\n-\nIIIIII\n"
        data, batch, chunk = b"".join(recs), 64, 16384
    elif case == "truncated":
        data = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(500)) + b"@last\nACG"
        batch, chunk = 50, 4096
    else:
        data = b"@a\nACGT\n+\nII I\n@b\nAC\n+\nII\n"
        check = 1
    path = tmp_path / "in.fastq"
    path.write_bytes(data)
    r = subprocess.run([exe, str(path), str(batch), str(chunk), str(check)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    if case == "validated_bad_quality":
        from oracle import oracle as O
        # the driver uses the generic schema; a space (32) is below its lower bound 33
        assert b"status=5" in r.stdout and b"Record number: 1" in r.stdout
        return
    assert r.stdout == _expect(data, batch, bool(check))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["clean_multi_chunk", "error_late", "ascii", "no_trailing_newline"])
def test_fasta_driver_reproduces_the_oracle(case, tmp_path):
    """tests/c_driver/bzq_facat.c: bzq_fasta_create -> bzq_fasta_ingest_open / _next -> bzq_fasta_copy_to_host ->
    bzq_fasta_format_error from plain C."""
    _build()
    exe = os.path.join(DRV, "bzq_facat")
    from oracle import fasta as F
    from fasta_fuzz import rand_fasta
    rng = np.random.default_rng(99)
    check, chunk = 0, 1 << 16
    if case == "clean_multi_chunk":
        data = rand_fasta(rng, 4000, 70, 5, dirty=0.0, lead_blank=1)
    elif case == "error_late":
        data = rand_fasta(rng, 3000, 60, 4, dirty=0.0) + b">empty\n>next\nAC\n"
    elif case == "ascii":
        data = rand_fasta(rng, 1500, 60, 4, dirty=0.0) + b">bad\nAC\x80GT\n>after\nAC\n"
        check = 1
    else:
        data = rand_fasta(rng, 2000, 60, 4, dirty=0.0, tail_newline=False, crlf=True)
    path = tmp_path / "in.fasta"
    path.write_bytes(data)
    r = subprocess.run([exe, str(path), str(chunk), str(check)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    w = F.flat_parse(data, bool(check))
    lines = r.stdout.split(b"\n")
    recs = [tuple(l.split(b"\t")) for l in lines if l and not l.startswith(b"#") and b"\t" in l]
    assert recs == [(i, s) for i, s in w.records()]
    assert (b"# records=%d " % w.n_records) in r.stdout and (b"status=%d" % w.status) in r.stdout
    if w.status != F.EOF:
        assert r.stdout.split(b"# error: ", 1)[1].rstrip(b"\n").decode("latin-1") == w.message
