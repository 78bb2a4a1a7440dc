"""Long randomized parity campaign of the HIP path against the oracle (run by hand on a GPU box; not collected by pytest):
    python tests/fuzz_campaign.py [--seeds 0:2000] [--seconds 300]
Varies everything the kernels branch on: record count, line lengths around 15/16/17 bytes and around the 16 KiB tile
edge, lines of several tiles, tiny records at the serial-path threshold (~1020 newlines per tile), dirt, tails, CRLF,
validation, offsets, batch size, is_eof.  Prints the first mismatch (seed + parameters) and exits 1."""
import argparse, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from fastq_fuzz import rand_stream, rand_record
from gpu_util import make_pair, check_against_oracle, check_views_against_oracle, EXPERIMENTS


def make_stream(rng):
    mode = int(rng.integers(0, 7))
    if mode == 0:      # generic fuzz
        return rand_stream(rng, n_records=int(rng.integers(1, 1500)), max_len=int(rng.choice([5, 40, 200, 1000])),
                           dirty=float(rng.choice([0, 0, 0.01, 0.05])), crlf=bool(rng.random() < 0.1)), "generic"
    if mode == 1:      # lengths around the 16-byte piece size
        L = int(rng.integers(13, 20))
        recs = [b"@" + bytes(rng.integers(48, 123, int(rng.integers(0, 20))).astype(np.uint8)) + b"\n" + b"ACGT" * 5 + b"\n+\n" + b"I" * 20 + b"\n" for _ in range(10)]
        parts = []
        for i in range(int(rng.integers(100, 3000))):
            l = L + int(rng.integers(-2, 3))
            s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), l))
            q = bytes(rng.integers(33, 127, l).astype(np.uint8))
            parts.append(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
        return b"".join(recs + parts), "piece-sized"
    if mode == 2:      # tiny records around the serial-path threshold (4 newlines per 16..17 bytes -> ~1000 per tile)
        parts = []
        for i in range(int(rng.integers(2000, 12000))):
            l = int(rng.integers(0, 5))
            parts.append(b"@" + (b"x" * int(rng.integers(0, 3))) + b"\n" + b"A" * l + b"\n+\n" + b"!" * l + b"\n")
        return b"".join(parts), "tiny"
    if mode == 3:      # lines of several tiles
        parts = []
        for i in range(int(rng.integers(3, 40))):
            l = int(rng.choice([16370, 16384, 16400, 33000, 50000, 100, 1]))
            s = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), l))
            q = bytes(rng.integers(33, 127, l).astype(np.uint8))
            parts.append(b"@long%d some description\n" % i + s + b"\n+\n" + q + b"\n")
        return b"".join(parts), "long-lines"
    if mode == 4:      # records placed so that every line role lands on a tile edge at some point
        l = int(rng.integers(100, 300))
        n = int(rng.integers(300, 1500))
        pad = b"@" + b"p" * int(rng.integers(0, 40)) + b"\nA\n+\n!\n"
        body = b"".join(b"@read%d\n" % i + b"A" * l + b"\n+\n" + b"I" * l + b"\n" for i in range(n))
        return pad + body, "edge-walk"
    if mode == 5:      # space runs in ids across edges
        parts = []
        for i in range(int(rng.integers(200, 2000))):
            sp = bytes(rng.choice([9, 11, 12, 13, 28, 29, 30, 32], int(rng.integers(0, 30))).astype(np.uint8))
            sp2 = bytes(rng.choice([9, 11, 12, 13, 28, 29, 30, 32], int(rng.integers(0, 30))).astype(np.uint8))
            l = int(rng.integers(0, 60))
            parts.append(b"@" + sp + b"id%d" % i + sp2 + b"\n" + b"C" * l + b"\n+\n" + b"#" * l + b"\n")
        return b"".join(parts), "spaces"
    # mode 6: mixture with one structural error somewhere
    data = bytearray(rand_stream(rng, n_records=int(rng.integers(50, 2000)), max_len=150, dirty=0.0, tail=0))
    if len(data):
        data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
    return bytes(data), "one-flip"


ap = argparse.ArgumentParser()
ap.add_argument("--seeds", default="0:100000")
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--views", action="store_true", help="views mode (ParserConfig.views_only) instead of the batch path")
args = ap.parse_args()
lo, hi = (int(x) for x in args.seeds.split(":"))
t0, done = time.time(), 0
pairs = {}
for seed in range(lo, hi):
    if time.time() - t0 > args.seconds:
        break
    rng = np.random.default_rng(seed)
    data, kind = make_stream(rng)
    kw = {}
    if rng.random() < 0.4:
        kw.update(check_ascii=True, check_quality=bool(rng.random() < 0.7))
        if rng.random() < 0.5:
            kw["quality_schema"] = str(rng.choice(["sanger", "solexa", "illumina_1.3", "illumina_1.5", "illumina_1.8"]))
    if rng.random() < 0.3 and not args.views:
        kw["emit_offsets"] = True
    if args.views:
        kw["views_only"] = True
    if rng.random() < 0.2:
        kw["buffer_capacity"] = int(rng.choice([64, 256, 4096, 65536]))
    if rng.random() < 0.15:
        kw["compat_simd_width"] = int(rng.choice([16, 32, 64]))
    bs = int(rng.choice([1, 7, 100, 256, 300, 4096]))   # (>= 256: the emit writes the per-batch ends itself, FusedArgs::fold)
    sp = [False, True, "v1"][int(rng.integers(0, 3))] if (EXPERIMENTS and rng.random() < 0.3 and not args.views) else False
    key = (bs, sp, tuple(sorted(kw.items())))
    if key not in pairs:
        if len(pairs) > 40:
            for c, _ in pairs.values(): c.close()
            pairs.clear()
        pairs[key] = make_pair(batch_size=bs, single_pass=sp, **kw)
    ctx, ocfg = pairs[key]
    is_eof = bool(rng.random() < 0.85)
    try:
        if args.views:
            check_views_against_oracle(ctx, ocfg, data, is_eof=is_eof, what=f"seed {seed} {kind}")
        else:
            check_against_oracle(ctx, ocfg, data, is_eof=is_eof, offsets=bool(kw.get("emit_offsets")), what=f"seed {seed} {kind}")
    except AssertionError as e:
        print(f"MISMATCH seed={seed} kind={kind} n={len(data)} bs={bs} single_pass={sp} is_eof={is_eof} kw={kw}\n{str(e)[:2000]}")
        sys.exit(1)
    done += 1
print(f"fuzz campaign: {done} streams in {time.time()-t0:.0f} s, all bit-identical to the oracle")
