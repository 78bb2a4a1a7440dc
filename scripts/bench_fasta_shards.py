"""What the byte-range shard protocol costs the FASTA parser on one GPU: the rank's range is cut at arbitrary offsets out of
benchmark/fasta-parser's synthetic stream (200-3800 bp, wrapped at 60), so the rank skips its head, probes its edges and
parses from its first header line.  No communicator (one rank); the exchange itself is bytes to kilobytes.
    python scripts/bench_fasta_shards.py [--records 1500000] [--steps 10]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blazeseq_amd as B

ap = argparse.ArgumentParser()
ap.add_argument("--records", type=int, default=1_500_000)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
fa = B.FastaContext(B.FastaParserConfig())
buf = fa.generate_synthetic_device(args.records, 200, 3800, 60)
n = buf.numel()
lo, hi = 1_000_003, n - 777_777          # inside records, not aligned to anything
ptr = int(buf.data_ptr())


def timed(f):
    for _ in range(3):
        r = f()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = f()
    return (time.perf_counter() - t0) / args.steps, r


t_whole, r0 = timed(lambda: fa.parse(ptr, n, True))
t_probe, s = timed(lambda: fa.shard_scan(ptr + lo, hi - lo))
head = int(s.first_header)   # lead_kind 0: the range starts inside a sequence line, its first header line is the rank's cut


def as_rank():   # what bzq_fasta_shard_stitch does on a middle rank, minus the two 64-byte all-gathers
    fa.shard_scan(ptr + lo, hi - lo)
    return fa.parse(ptr + lo + head, hi - lo - head, True)


t_stitch, r1 = timed(as_rank)
print(f"whole stream   {n/1e9:.2f} GB  {t_whole*1e3:7.3f} ms  {n/t_whole/1e9:7.1f} GB/s  {int(r0.n_records)} records")
print(f"probe          first_header={int(s.first_header)} lead_kind={int(s.lead_kind)} tail_open={int(s.tail_open)}  {t_probe*1e3:7.3f} ms")
print(f"shard stitch   {(hi-lo)/1e9:.2f} GB  {t_stitch*1e3:7.3f} ms  {(hi-lo)/t_stitch/1e9:7.1f} GB/s  {int(r1.n_records)} records, "
      f"head {head} B")
