"""FASTA over byte-range shards on the GPU: k_fa_probe_headers / k_fa_probe_edges against the byte-by-byte restatement,
and bzq_fasta_shard_stitch -- one plain-C process per rank (tests/c_driver/bzq_fasta_shard.c), all on GPU 0 through the
host (shm) transport -- against the oracle's sequential parse of the WHOLE stream: records, status, error text."""
import os
import subprocess

import numpy as np
import pytest

from oracle import fasta as FO
from tests.fasta_fuzz import rand_fasta, rand_soup
from tests.fasta_shard_model import summary_of

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DRV = os.path.join(HERE, "c_driver")
CAP = 32768   # the smallest line capacity the device parser takes


def _exe():
    subprocess.run(["make", "-C", DRV], check=True, capture_output=True)
    return os.path.join(DRV, "bzq_fasta_shard")


def _device_summary(fa, ctx, piece: np.ndarray):
    from blazeseq_amd import _lib as L
    import ctypes as C
    d = C.c_void_p()
    assert L.lib().bzq_device_alloc(ctx.h, max(piece.size, 1) + 64, C.byref(d)) == 0
    try:
        if piece.size:
            assert L.lib().bzq_copy_to_device(ctx.h, d, piece.ctypes.data, piece.size) == 0
        s = fa.shard_scan(d.value, piece.size)
        return (int(s.n_bytes), int(s.first_header), int(s.lead_kind), int(s.last_byte), int(s.tail_open))
    finally:
        L.lib().bzq_device_free(ctx.h, d)


def test_probe_kernels_match_the_restatement():
    from blazeseq_amd.fasta import FastaContext
    from blazeseq_amd.parser import Context
    fa, ctx = FastaContext(), Context()
    rng = np.random.default_rng(3)
    pieces = [b"", b" ", b"\n", b">", b"  >x", b"\n  ", b"a\n  >b", b"\n\t\x1c >q\n", b"ACGT" * 10, b" " * 5000, b" " * 5000 + b">z",
              b"x\n" + b" " * 40000 + b">far\nA\n", b"x\n" + b" " * 40000, b"x\n" + b" " * 300000 + b">too far\nA\n", b"A" * 70000 + b"\n>late\nC", b">a\n" + b"AC\n" * 30000 + b"  "]
    for _ in range(60):
        k = int(rng.integers(0, 3))
        if k == 0:
            d = rand_fasta(rng, n_records=int(rng.integers(1, 40)), dirty=0.2, max_line=int(rng.integers(5, 200)))
        elif k == 1:
            d = rand_soup(rng, int(rng.integers(1, 60000)))
        else:
            d = rand_soup(rng, int(rng.integers(1, 60000)), weights=[0.02, 0.05, 8, 2, 1, 1, 0.5, 0.1, 0.1])   # space runs over tile edges
        a, b = sorted(int(x) for x in rng.integers(0, len(d) + 1, 2))
        pieces.append(d[a:b])
    for p in pieces:
        a = np.frombuffer(p, dtype=np.uint8)
        assert _device_summary(fa, ctx, a) == summary_of(a, walk_cap=256 * 1024), p[:80]
    # the scans give up after line_capacity bytes of spaces (a line that long fails wherever it is parsed)
    from blazeseq_amd.fasta import FastaParserConfig
    small = FastaContext(FastaParserConfig(line_capacity=32768))
    for p in (b" " * 40000 + b">a\nC\n", b"x\n" + b" " * 40000, b"x\n" + b" " * 40000 + b">far\nA\n", b" " * 32768 + b"\n>b\nA", b"AC\n" + b" " * 36863,
              b"AC\n" + b" " * 36864, b"AC\n" + b" " * 36865, b" " * 36863 + b"G", b" " * 36864 + b"G"):
        a = np.frombuffer(p, dtype=np.uint8)
        assert _device_summary(small, ctx, a) == summary_of(a, walk_cap=32768), (len(p), p[:20])


def _run(data: bytes, cuts, tmp_path, check=False, cap=0, name="s"):
    exe = _exe()
    path = tmp_path / f"{name}.fa"
    path.write_bytes(data)
    bounds = [0, *cuts, len(data)]
    P = len(bounds) - 1
    shm = f"fa{os.getpid()}_{name}"
    procs = [subprocess.Popen([exe, "shm", str(r), str(P), shm, str(path), str(bounds[r]), str(bounds[r + 1]), str(int(check)), str(cap)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(P)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e.decode()
        outs.append(o)
    return outs


def _unhex(x: bytes) -> bytes:
    return b"" if x == b"-" else bytes.fromhex(x.decode())


def _check(outs, data: bytes, check=False, cap=0):
    whole = FO.flat_parse(np.frombuffer(data, dtype=np.uint8), check_ascii=check, line_cap=cap or FO.DEFAULT_CAPACITY)
    recs, msg, before = [], None, 0
    for r, o in enumerate(outs):
        lines = o.split(b"\n")
        trailer = next(l for l in lines if l.startswith(b"# rank="))
        kv = dict(x.split(b"=") for x in trailer[2:].split())
        mine = [tuple(_unhex(x) for x in l.split(b" ")) for l in lines if l and not l.startswith(b"#")]
        assert int(kv[b"records"]) == len(mine) and int(kv[b"before"]) == before, (r, trailer)
        before += len(mine)
        recs += mine
        assert int(kv[b"global_records"]) == whole.n_records, (r, trailer)
        assert int(kv[b"stream_status"]) == whole.status, (r, trailer, whole.message)
        assert int(kv[b"first_error"]) == (-1 if whole.status == 6 else whole.n_records), (r, trailer)
        for l in lines:
            if l.startswith(b"# error "):
                assert int(kv[b"error_rank"]) == r
                msg = _unhex(l[8:]).decode("latin-1")
    assert recs == whole.records()
    if whole.status != 6:
        assert msg == whole.message


def test_small_streams_every_cut(tmp_path):
    data = b"\n  \n>a 1\nACGT\nAC\n \t>b\nTT\n\n>c\n  G  \n>d x\nA"
    for c in range(0, len(data) + 1, 3):
        _check(_run(data, [c], tmp_path, name=f"c{c}"), data)
    for i, (c, d) in enumerate([(0, 0), (3, 3), (5, 30), (17, 19), (18, 18), (len(data), len(data)), (1, 2)]):
        _check(_run(data, [c, d], tmp_path, name=f"d{i}"), data)


@pytest.mark.parametrize("seed", range(12))
def test_random_streams_random_cuts(seed, tmp_path):
    rng = np.random.default_rng(100 + seed)
    k = seed % 3
    if k == 0:
        data = rand_fasta(rng, n_records=int(rng.integers(50, 600)), max_line=int(rng.integers(20, 300)), crlf=bool(rng.random() < 0.3),
                          tail_newline=bool(rng.random() < 0.7))
    elif k == 1:
        data = rand_fasta(rng, n_records=int(rng.integers(50, 600)), dirty=0.02, lead_blank=int(rng.integers(0, 3)))
    else:
        data = rand_soup(rng, int(rng.integers(100, 50000)), weights=[0.3, 4, 3, 1, 6, 6, 1, 0.02, 0.02])
    check = bool(rng.random() < 0.5)
    for j in range(3):
        P = int(rng.integers(2, 5))
        cuts = sorted(int(x) for x in rng.integers(0, len(data) + 1, P - 1))
        _check(_run(data, cuts, tmp_path, check=check, name=f"r{j}"), data, check=check)


def test_record_longer_than_several_shards(tmp_path):
    rng = np.random.default_rng(5)
    big = b"\n".join(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 70)) for _ in range(9000))   # ~640 KB in one record
    data = b">small\nACGT\n>big one\n" + big + b"\n>after\nTTTT\n  >last\nG"
    n = len(data)
    _check(_run(data, [n // 5, 2 * n // 5, 3 * n // 5, 4 * n // 5], tmp_path, name="big5"), data)
    _check(_run(data, [12, n - 20, n - 9], tmp_path, name="edge"), data)


def test_order_of_events_across_a_cut(tmp_path):
    long_hdr = b">" + b"h" * (CAP + 10)
    cases = [
        b">a\nAC\n>b\nGG\n" + long_hdr + b"\nA\n>z\nT\n",           # the record before a too-long header line is lost
        b">a\nAC\n>b\n" + long_hdr + b"\nA\n",                       # ... and its empty sequence is never reported
        b">a\nAC\n>b\n\x80\n" + long_hdr + b"\nA\n",                 # ... nor its non-ASCII byte
        b">a\nAC\n>b\nGG\n" + b"C" * (CAP + 5) + b"\n>c\nA\n",       # a too-long sequence line: the open record is its own
        b">a\nAC\n>b\n>c\nA\n>d\nTT\n",                              # empty sequence, numbers in the text are stream-global
        b"\n\n>a\nAC\n>b\nG\x80G\n>c\nA\n",
        b"ACGT\n>a\nA\n",
    ]
    for i, data in enumerate(cases):
        marks = [data.find(b">b"), data.find(long_hdr) if long_hdr in data else data.find(b">c"), len(data) - 3]
        for j, cuts in enumerate([[marks[1]], [marks[0], marks[1]], [marks[0] + 1, marks[1] + 1], [marks[1] - 1, marks[2]], [2, 4, marks[1]]]):
            cuts = sorted(max(0, c) for c in cuts)
            _check(_run(data, cuts, tmp_path, check=True, cap=CAP, name=f"o{i}_{j}"), data, check=True, cap=CAP)


def test_world_1_over_rccl_and_without_a_communicator(tmp_path):
    from blazeseq_amd.fasta import FastaContext
    from blazeseq_amd.parser import Context
    from blazeseq_amd import _lib as L
    import ctypes as C
    rng = np.random.default_rng(9)
    data = rand_fasta(rng, n_records=300, dirty=0.0)
    path = tmp_path / "w1.fa"
    path.write_bytes(data)
    p = subprocess.run([_exe(), "rccl", "0", "1", str(tmp_path / "id"), str(path), "0", str(len(data))], capture_output=True, timeout=180)
    assert p.returncode == 0, p.stderr.decode()
    out = b"\n".join(l for l in p.stdout.split(b"\n") if l.startswith(b"#") or l.count(b" ") == 1)   # RCCL prints a banner
    _check([out], data)
    # no communicator at all: comm_ctx = None
    fa, ctx = FastaContext(), Context()
    a = np.frombuffer(data, dtype=np.uint8)
    d = C.c_void_p()
    assert L.lib().bzq_device_alloc(ctx.h, a.size + 64, C.byref(d)) == 0
    assert L.lib().bzq_copy_to_device(ctx.h, d, a.ctypes.data, a.size) == 0
    res = fa.shard_stitch(None, d.value, a.size, a.size + 64)
    whole = FO.flat_parse(a)
    assert int(res.chunk.n_records) == whole.n_records == int(res.global_records) and res.stream_status == 6
    ids, id_ends, seq, seq_ends, _ = fa.columns(res.chunk)
    np.testing.assert_array_equal(seq, whole.seq_bytes)
    np.testing.assert_array_equal(id_ends, whole.id_ends)
    L.lib().bzq_device_free(ctx.h, d)


def test_a_head_that_does_not_fit_fails_on_every_rank_instead_of_hanging(tmp_path):
    """4 MiB of room behind every range (and 4 MiB halo slots in the shm transport): a record whose continuation in the
    following ranges is longer fails the call on ALL ranks before anything is exchanged (a rank bailing out alone would leave
    its peers waiting for ever)."""
    exe = _exe()
    data = b">a\nAC\n>long\n" + (b"ACGTACGTAC" * 6 + b"\n") * 160000 + b">z\nT\n"   # ~9.8 MB in one record
    path = tmp_path / "long.fa"
    path.write_bytes(data)
    n = len(data)
    bounds = [0, 1000, 6_000_000, n]   # rank 1's 6 MB are all the head of rank 0's record
    shm = f"fa{os.getpid()}_cap"
    procs = [subprocess.Popen([exe, "shm", str(r), "3", shm, str(path), str(bounds[r]), str(bounds[r + 1])],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(3)]
    for p in procs:
        try:
            _, e = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        # (the driver leaves 4 MiB behind the range: the room check of the plan fires before the transport's own limit)
        assert p.returncode == 4 and (b"no room for its halo" in e or b"exceeds the halo capacity" in e), (p.returncode, e[-300:])
