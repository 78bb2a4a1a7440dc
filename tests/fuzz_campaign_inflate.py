"""Device BGZF inflate, randomized campaign (by hand on a GPU box; 15 s under the driver): batches of blocks of random content
(FASTQ, FASTA-like, random bytes over alphabets of 1..256 symbols, runs, periodic data, text), size 0..65536, zlib level 0..9,
every strategy (default, filtered, Huffman only, RLE, fixed) and memLevel 1..9 -- what the device decodes == what was compressed.
    python tests/fuzz_campaign_inflate.py [--seconds 120]"""
import argparse, os, sys, time, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from fastq_fuzz import rand_stream
from fasta_fuzz import rand_fasta
from bgzf_util import bgzf_block
from test_gpu_bgzf_inflate import inflate_on_device
from blazeseq_amd.parser import Context

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
args = ap.parse_args()
ctx = Context()
t0, seed, blocks, nbytes = time.time(), 90_000, 0, 0
kinds = {}
while time.time() - t0 < args.seconds:
    seed += 1
    rng = np.random.default_rng(seed)
    parts, want = [], []
    for _ in range(int(rng.integers(1, 40))):
        kind = int(rng.integers(0, 8))
        n = int(rng.choice([0, 1, 2, 100, 5000, 30000, 65000, 65536])) if rng.random() < 0.3 else int(rng.integers(0, 65537))
        if kind == 0:
            piece = rand_stream(rng, n_records=max(1, n // 300), max_len=int(rng.choice([50, 150, 5000])), dirty=0.0, tail=0)
        elif kind == 1:
            piece = rand_fasta(rng, n_records=max(1, n // 400), max_line=int(rng.choice([60, 70, 200])))
        elif kind == 2:
            piece = rng.integers(0, int(rng.integers(1, 257)), n, dtype=np.uint8).tobytes()
        elif kind == 3:
            piece = bytes([int(rng.integers(0, 256))]) * n
        elif kind == 4:
            piece = bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)) * (n // 2 + 1)
        elif kind == 5:
            piece = bytes(rng.integers(0, 256, int(rng.integers(1000, 33000)), dtype=np.uint8)) * 70
        elif kind == 6:
            words = [bytes(rng.integers(97, 123, int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(int(rng.integers(2, 200)))]
            piece = b" ".join(words[int(i)] for i in rng.integers(0, len(words), n // 4 + 1))
        else:
            a = rng.integers(0, 256, n, dtype=np.uint8); a[rng.random(n) < 0.9] = 65
            piece = a.tobytes()
        piece = piece[:n]
        level = int(rng.integers(0, 10))
        strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        while True:
            try:
                parts.append(bgzf_block(piece, level, strat, int(rng.integers(1, 10))))
                break
            except AssertionError:   # incompressible at this level: a smaller piece
                piece = piece[: len(piece) * 3 // 4]
        want.append(piece)
        kinds[kind] = kinds.get(kind, 0) + 1
    got = inflate_on_device(ctx, b"".join(parts))
    if got != b"".join(want):
        print(f"MISMATCH seed={seed}")
        sys.exit(1)
    blocks += len(parts); nbytes += sum(len(w) for w in want)
print(f"inflate campaign: {blocks} blocks ({nbytes/1e6:.0f} MB) identical to the bytes zlib compressed, by kind {dict(sorted(kinds.items()))} in {time.time()-t0:.0f} s")
