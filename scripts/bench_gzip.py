"""Plain (non-BGZF) gzip files through the device decoder (bzq_gzip.hpp) and end to end through the ingest.
Files: the benchmark's synthetic 150 bp FASTQ, --gb of it, as
  * "single": ONE gzip member, one zlib stream of the given level (what `gzip -N reads.fastq` writes);
  * "pigz":   ONE member whose DEFLATE stream was compressed in independent 16 MiB pieces on all host cores (what `pigz -i` writes);
  * "multi":  a member per 48 MB (what `cat a.gz b.gz ...` gives).
Every decode is compared with the FASTQ it was made from, byte for byte, on the device.
    python scripts/bench_gzip.py [--gb 1.0] [--levels 1,6,9] [--dir /dev/shm]"""
import argparse, os, struct, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import blazeseq_amd as B
from blazeseq_amd import _lib as L
from tests.gzip_util import gzip_member, sequencer_like

ap = argparse.ArgumentParser()
ap.add_argument("--gb", type=float, default=1.0)
ap.add_argument("--levels", default="1,6,9")
ap.add_argument("--kinds", default="single,pigz,multi")
ap.add_argument("--slice-mb", type=int, default=48)
ap.add_argument("--piece-mib", type=int, default=256, help="compressed bytes handed to the decoder per call")
ap.add_argument("--chunk-kib", type=int, default=16, help="compressed bytes per decoder wave")
ap.add_argument("--chunk-mib", type=int, default=256, help="ingest chunk (decompressed bytes)")
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--no-stage", action="store_true", help="no bzq_gzip_stage read-ahead of the next piece")
ap.add_argument("--no-ingest", action="store_true")
ap.add_argument("--data", default="synthetic", choices=["synthetic", "sequencer"], help="synthetic: the benchmark's generator (random qualities); sequencer: quality runs, duplicates, poly-G tails")
args = ap.parse_args()

ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)


if args.data == "sequencer":
    plain, n_rec = sequencer_like(args.slice_mb << 20)
    d_plain = torch.from_numpy(np.frombuffer(plain, dtype=np.uint8).copy()).cuda()
else:
    n_rec = args.slice_mb * (1 << 20) // 318
    size = ctx.generate_synthetic_device(n_rec, 150, 33, 73, "generic", 0, 0)
    buf = torch.empty(size + 64, dtype=torch.uint8, device="cuda")
    ctx.generate_synthetic_device(n_rec, 150, 33, 73, "generic", buf.data_ptr(), buf.numel())
    torch.cuda.synchronize()
    d_plain = buf[:size]
    plain = d_plain.cpu().numpy().tobytes()
reps = max(1, int(args.gb * 1e9 / len(plain)))
total_plain, total_rec = len(plain) * reps, n_rec * reps
HDR = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3])


def trailer():
    crc = 0
    for _ in range(reps):
        crc = zlib.crc32(plain, crc)
    return struct.pack("<II", crc & 0xFFFFFFFF, total_plain & 0xFFFFFFFF)


def build(kind, level):
    t0 = time.perf_counter()
    if kind == "multi":
        m = gzip_member(plain, level)
        comp = m * reps
    elif kind == "single":
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = HDR + b"".join(co.compress(plain) for _ in range(reps)) + co.flush() + trailer()
    else:   # pigz -i: independent pieces, each ending on a sync flush; the last one finishes the stream
        P = 16 << 20
        pieces = [plain[i:i + P] for i in range(0, len(plain), P)]

        def one(a):
            data, last = a
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            return co.compress(data) + (co.flush() if last else co.flush(zlib.Z_SYNC_FLUSH))
        with ThreadPoolExecutor(32) as ex:
            mid = list(ex.map(one, [(p, False) for p in pieces]))
            fin = list(ex.map(one, [(p, i + 1 == len(pieces)) for i, p in enumerate(pieces)]))
        comp = HDR + b"".join(mid) * (reps - 1) + b"".join(fin) + trailer()
    return comp, time.perf_counter() - t0


def check(out_t, n):
    """out_t[:n] == the FASTQ repeated, compared on the device"""
    assert n == total_plain, (n, total_plain)
    v = out_t[:n].view(reps, len(plain))
    assert bool((v == d_plain.unsqueeze(0)).all()), "device output differs from the FASTQ the file was made from"


out = torch.empty(total_plain + (1 << 20), dtype=torch.uint8, device="cuda")
lib = L.lib()
for kind in args.kinds.split(","):
    for level in [int(x) for x in args.levels.split(",")]:
        comp, t_build = build(kind, level)
        print(f"\n{kind} level {level}: {len(comp)/1e9:.3f} GB compressed ({total_plain/len(comp):.2f}x) = {total_plain/1e9:.2f} GB of FASTQ  [built in {t_build:.0f} s]", flush=True)
        # ---- the decoder alone: pinned host memory -> device bytes
        pin = torch.from_numpy(np.frombuffer(comp, dtype=np.uint8).copy()).pin_memory()
        best = None
        for rep in range(3):
            dec = B.GzipDecoder(ctx, args.chunk_kib << 10)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            off, got = 0, 0
            piece = args.piece_mib << 20
            if not args.no_stage:
                dec.stage(pin[0:piece].numpy())
            while off < pin.numel() or not dec.finished:
                part = pin[off:off + piece].numpy()
                off += part.size
                if not args.no_stage and off < pin.numel():
                    dec.stage(pin[off:off + piece].numpy())   # the next piece travels while this one is decoded
                while True:
                    nb, more = dec.feed(part, off >= pin.numel(), out.data_ptr() + got, out.numel() - got)
                    got += nb
                    if not more:
                        break
                    part = part[:0]
                if off >= pin.numel():
                    break
            dt = time.perf_counter() - t1
            st = dec.stats()
            dec.close()
            if rep == 0:
                check(out, got)
            best = dt if best is None or dt < best else best
        print(f"  decoder (pinned host -> device, {args.piece_mib} MiB pieces, {args.chunk_kib} KiB per wave): {total_plain/best/1e9:6.2f} GB/s of FASTQ "
              f"({best*1e3:.0f} ms; {st.chain_jobs} of {st.chunks_with_start + st.pieces} decoder runs in the output, {st.fallback_jobs} restarts, {st.members} members, {st.pool_retries} pool retries)", flush=True)
        del pin
        if args.no_ingest:
            continue
        # ---- end to end: file (page cache) -> ingest -> parser
        path = os.path.join(args.dir if os.path.isdir(args.dir) else "/tmp", "bzq_gzip_bench.fastq.gz")
        with open(path, "wb") as f:
            f.write(comp)
        for gpu in (1, 0):
            if gpu == 0 and (kind != "single" or level != 6):
                continue   # the host zlib path once: it is slow
            c = B.Context(B.ParserConfig(), "generic", 4096, 0)
            c.set_option("ingest_gpu_inflate", gpu)
            best = None
            for rep in range(2 if gpu else 1):
                t1 = time.perf_counter()
                ing = B.Ingest(c, path, chunk_bytes=args.chunk_mib << 20, n_threads=8)
                taken, total = 0, 0
                while True:
                    res = ing.next(taken)
                    taken = int(res.n_records); total += taken
                    if int(res.status) != L.OK:
                        break
                dt = time.perf_counter() - t1
                ist = ing.stats()
                ing.close()
                assert total == total_rec and int(res.status) == L.EOF, (total, total_rec, res.status)
                if best is None or dt < best:
                    best, how = dt, f"producer busy {ist.read_s*1e3:.0f} ms, consumer waiting {ist.wait_s*1e3:.0f} ms; the rest is the open"
            print(f"  file -> records, inflate {'on the device' if gpu else 'by zlib gzread on the host (the GZFile way)'}: {total_plain/best/1e9:6.2f} GB/s of FASTQ end to end ({best*1e3:.0f} ms: {how})", flush=True)
        os.remove(path)
