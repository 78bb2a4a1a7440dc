"""Parser-level randomized campaign (by hand on a GPU box): blazeseq_amd.FastqParser over random chunk sizes -- from
memory and from files through the native ingest (plain / gzip / BGZF) -- against the oracle's streaming parser: same
batches, same terminal event.   python tests/fuzz_campaign_parser.py [--seconds 240]"""
import argparse, gzip, os, struct, sys, tempfile, time, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from fastq_fuzz import rand_stream
from oracle import oracle as O
import blazeseq_amd as B


def bgzf(data, block=65280):
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1)
                   + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def run_ref(data, bs, kw):
    sp = O.StreamParser(np.frombuffer(data, dtype=np.uint8), O.make_config(batch_size=bs, **kw))
    out, err = [], None
    try:
        while True:
            b = sp.next_batch(bs)
            if len(b) == 0:
                break
            out.append((b.ends, b.id_ends, b.seq_bytes, b.qual_bytes, b.id_bytes))
    except O.OracleError as e:
        err = str(e)
    return out, err


def run_gpu(src, bs, kw, chunk):
    p = B.FastqParser(src, batch_size=bs, chunk_bytes=chunk, config=B.ParserConfig(**kw), reader_threads=2)
    out, err = [], None
    try:
        while True:
            b = p.next_batch(bs)
            if len(b) == 0:
                break
            out.append((b._ends.tolist(), b._id_ends.tolist(), b._sequence_bytes.tobytes(), b._quality_bytes.tobytes(), b._id_bytes.tobytes()))
    except B.ParseError as e:
        err = e.message.decode("latin-1")
    return out, err


ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=240)
ap.add_argument("--seed0", type=int, default=0)
args = ap.parse_args()
t0, done = time.time(), 0
tmp = tempfile.mkdtemp()
seed = args.seed0
while time.time() - t0 < args.seconds:
    seed += 1
    rng = np.random.default_rng(seed)
    tail = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5]))
    data = rand_stream(rng, n_records=int(rng.integers(1, 4000)), max_len=int(rng.choice([10, 60, 150, 600])),
                       dirty=float(rng.choice([0, 0, 0, 0.002])), tail=tail, crlf=bool(rng.random() < 0.1))
    kw = {}
    if rng.random() < 0.4:
        kw = dict(check_ascii=True, check_quality=bool(rng.random() < 0.6))
    bs = int(rng.choice([1, 5, 64, 1000, 4096]))
    if bs == 1 and len(data) > 60000:
        bs = 64
    chunk = int(rng.choice([1 << 16, 100_000, 1 << 20, 1 << 30]))
    ref = run_ref(data, bs, kw)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        src, name = data, "memory"
    else:
        path = os.path.join(tmp, f"f{seed}.fastq" + ("" if kind == 1 else ".gz"))
        with open(path, "wb") as f:
            f.write(data if kind == 1 else (gzip.compress(data, 1) if kind == 2 else bgzf(data)))
        src, name = path, ["", "file", "gzip", "bgzf"][kind]
    got = run_gpu(src, bs, kw, chunk)
    if kind:
        os.remove(path)
    if got != ref:
        multi_chunk = len(data) > chunk
        same_batches = got[0] == ref[0]
        print(f"MISMATCH seed={seed} src={name} n={len(data)} bs={bs} chunk={chunk} tail={tail} kw={kw}\n"
              f"  batches {len(got[0])} vs {len(ref[0])}, same={same_batches}\n  gpu err: {got[1]!r}\n  ref err: {ref[1]!r}")
        sys.exit(1)
    done += 1
print(f"parser campaign: {done} streams identical (batches, terminal code and error text) in {time.time()-t0:.0f} s")
