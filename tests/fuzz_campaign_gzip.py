"""Device gzip decode (bzq_gzip.hpp), randomized campaign (by hand on a GPU box; 15 s under the driver): gzip files of random
content (FASTQ, FASTA-like, random bytes over alphabets of 1..256 symbols, runs, periodic data at every distance, text), 0 ..
3 MB, 1..6 members, zlib level 0..9, every strategy (default, filtered, Huffman only, RLE, fixed) and memLevel 1..9, full
flushes inside members, header fields (name, comment, extra, header CRC), trailing garbage -- fed to the decoder in pieces of a
random size, with a random number of compressed bytes per decoder wave and an output buffer that may be far smaller than the
output: what the device delivers == the bytes zlib compressed.  Every tenth stream is damaged (bit flip / truncation / wrong
trailer) and must fail the call.
    python tests/fuzz_campaign_gzip.py [--seconds 120]"""
import argparse, os, sys, time, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from fastq_fuzz import rand_stream
from fasta_fuzz import rand_fasta
from gzip_util import DeviceGunzip, gzip_member
from blazeseq_amd.parser import Context

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
args = ap.parse_args()
ctx = Context()
t0, seed, streams, nbytes, refused = time.time(), 130_000, 0, 0, 0
device_alone = 0
kinds = {}


def content(rng, kind, n):
    if kind == 0:
        return rand_stream(rng, n_records=max(1, n // 300), max_len=int(rng.choice([50, 150, 5000])), dirty=0.0, tail=0)
    if kind == 1:
        return rand_fasta(rng, n_records=max(1, n // 400), max_line=int(rng.choice([60, 70, 200])))
    if kind == 2:
        return rng.integers(0, int(rng.integers(1, 257)), n, dtype=np.uint8).tobytes()
    if kind == 3:
        return bytes([int(rng.integers(0, 256))]) * n
    if kind == 4:
        return bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)) * (n // 2 + 1)
    if kind == 5:   # matches at distances up to the whole window, across whatever cuts the decoder makes
        return bytes(rng.integers(0, 256, int(rng.integers(1000, 33000)), dtype=np.uint8)) * (n // 1000 + 2)
    if kind == 6:
        words = [bytes(rng.integers(97, 123, int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(int(rng.integers(2, 200)))]
        return b" ".join(words[int(i)] for i in rng.integers(0, len(words), n // 4 + 1))
    a = rng.integers(0, 256, n, dtype=np.uint8); a[rng.random(n) < 0.9] = 65
    return a.tobytes()


while time.time() - t0 < args.seconds:
    seed += 1
    rng = np.random.default_rng(seed)
    parts, comp = [], []
    for _ in range(int(rng.integers(1, 7))):
        kind = int(rng.integers(0, 8))
        n = int(rng.choice([0, 1, 2, 100, 5000, 70000])) if rng.random() < 0.25 else int(rng.integers(0, 600000))
        piece = content(rng, kind, n)[:n]
        kw = {}
        if rng.random() < 0.3:
            kw = dict(name=bytes(rng.integers(1, 256, int(rng.integers(1, 400)), dtype=np.uint8)), comment=b"c" * int(rng.integers(0, 100)),
                      extra=bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8)), hcrc=bool(rng.random() < 0.5))
        comp.append(gzip_member(piece, int(rng.integers(0, 10)), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])),
                                mem_level=int(rng.integers(1, 10)), flush_every=int(rng.choice([0, 0, 0, 20000, 150000])), **kw))
        parts.append(piece)
        kinds[kind] = kinds.get(kind, 0) + 1
    data, blob = b"".join(parts), b"".join(comp)
    damaged = seed % 10 == 0 and len(blob) > 64
    if damaged:
        how = int(rng.integers(0, 3))
        b2 = bytearray(blob)
        if how == 0:
            b2[int(rng.integers(20, len(b2) - 8))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:
            b2 = b2[:int(rng.integers(1, len(b2) - 1))]
        else:
            b2[-int(rng.integers(1, 9))] ^= 0x40
        blob = bytes(b2)
    elif rng.random() < 0.2:
        blob += bytes(rng.integers(0, 256, int(rng.integers(1, 50)), dtype=np.uint8)).replace(b"\x1f\x8b", b"..")   # bytes behind the last member
    piece_size = int(rng.choice([0, 1 << 16, 1 << 20, int(rng.integers(1, len(blob) + 2))]))
    cap = len(data) + 4096 if rng.random() < 0.6 else max(300000, len(data) // int(rng.integers(2, 6)))
    g = DeviceGunzip(ctx, cap, chunk_bytes=int(rng.choice([4096, 8192, 16384, 32768, 65536])))
    # round 4: pieces staged ahead (the next piece's finder / decoders then run under this piece's last kernels), the finder behind
    # the piece's own copy, the hand-over of a stretch without findable block starts to the host and back -- in every combination
    ahead = int(rng.choice([0, 0, 1, 2, 3])) if piece_size else 0
    g.dec.set_option("predecode", int(rng.random() < 0.8))
    g.dec.set_option("early_find", int(rng.random() < 0.5))
    g.dec.set_option("chain_l2", int(rng.random() < 0.5))     # round 5: beside a predecode, the chain kernels that read their window through the L2
    hc = int(rng.random() < 0.5)   # half of the streams: the device alone (nothing of the result comes from zlib, the checker)
    g.dec.set_option("host_continuation", hc)
    g.dec.set_option("far_kib", int(rng.choice([16, 64, 256])))
    g.dec.set_option("host_budget_kib", int(rng.choice([64, 512, 32768])))
    try:
        got = g.decode(blob, piece_size, ahead=ahead)
        if damaged and got != data:
            # a cut that falls exactly behind a member leaves a valid, shorter file (and bytes behind it are ignored)
            prefixes = {b"".join(parts[:k]) for k in range(len(parts) + 1)}
            if not (how == 1 and got in prefixes):
                print(f"WRONG BYTES from a damaged stream instead of an error, seed={seed}")
                sys.exit(1)
            got = data
        if got != data:
            print(f"MISMATCH seed={seed} ({len(got)} vs {len(data)} bytes)")
            sys.exit(1)
        if hc == 0 and g.dec.set_option("host_calls", 0) != 0:
            print(f"HOST CALLS with host_continuation = 0, seed={seed}")
            sys.exit(1)
        device_alone += (hc == 0)
        # (a flipped bit can land in a header field or in bytes zlib ignores too: then the stream still decodes to the same bytes)
    except RuntimeError as e:
        if "out_capacity" in str(e) and cap < len(data) + 4096:
            # documented limit: the buffer must take the output of one DEFLATE block (up to 8 MB from zlib at memLevel 9);
            # this stream has one that the small buffer of this round does not -- it must still decode into a full-size one
            g.close()
            g = DeviceGunzip(ctx, len(data) + 4096, chunk_bytes=16384)
            try:
                assert g.decode(blob, piece_size) == data or damaged
            except RuntimeError:
                assert damaged
        elif not damaged:
            print(f"REFUSED a valid stream, seed={seed}: {e}")
            sys.exit(1)
        else:
            refused += 1
    g.close()
    streams += 1; nbytes += len(data)
print(f"gzip campaign: {streams} streams ({nbytes/1e6:.0f} MB; {device_alone} by the device alone; {refused} damaged ones refused) identical to the bytes zlib compressed, by kind {dict(sorted(kinds.items()))} in {time.time()-t0:.0f} s")
