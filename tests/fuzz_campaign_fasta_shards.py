"""FASTA byte-range shards, randomized campaign (by hand on a GPU box): a stream cut into 2..8 byte ranges ANYWHERE; the
device's probe kernels summarize every range, bzq_fasta_plan_shards places the cuts, the device parser takes every owner's
region (it starts wherever the header line starts: any alignment), the outcome rules of bzq_fasta_shard_stitch are applied
(tests/fasta_shard_model.py) -> records, status and error text == the oracle's sequential parse of the whole stream.
The transports and the C resolve code themselves run in tests/test_gpu_fasta_shards.py (one process per rank).
    python tests/fuzz_campaign_fasta_shards.py [--seconds 180]"""
import argparse, ctypes as C, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from fasta_fuzz import rand_fasta, rand_soup
from oracle import fasta as FO
from tests.fasta_shard_model import stitch, summary_of
from blazeseq_amd import _lib as L
from blazeseq_amd.fasta import FastaContext, FastaParserConfig
from blazeseq_amd.parser import Context

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=180)
args = ap.parse_args()
CAP = 32768
ctx = Context()
fas = {ca: FastaContext(FastaParserConfig(check_ascii=ca, line_capacity=CAP)) for ca in (False, True)}
lib = L.lib()
lib.bzq_fasta_error_open_record.restype = C.c_int64
lib.bzq_fasta_error_open_record.argtypes = [C.c_void_p]
d_buf = C.c_void_p()
DCAP = 8 << 20
assert lib.bzq_device_alloc(ctx.h, DCAP, C.byref(d_buf)) == 0


def upload(a: np.ndarray, off: int):
    if a.size:
        assert lib.bzq_copy_to_device(ctx.h, C.c_void_p(d_buf.value + off), a.ctypes.data, a.size) == 0
    return d_buf.value + off


class DevFlat:
    def __init__(self, fa, res):
        self.status, self.n_records = int(res.status), int(res.n_records)
        k = int(lib.bzq_fasta_error_open_record(fa._h))
        self.err_record = k if self.status != 6 else -1
        ids, id_ends, seq, seq_ends, _ = fa.columns(res)
        self._cols = (ids, id_ends, seq, seq_ends)
        self.message = fa.error_text().decode("latin-1") if self.status != 6 else ""

    def records(self):
        ids, ie, seq, se = self._cols
        out, s0, i0 = [], 0, 0
        for r in range(self.n_records):
            out.append((ids[i0:int(ie[r])].tobytes(), seq[s0:int(se[r])].tobytes()))
            s0, i0 = int(se[r]), int(ie[r])
        return out


t0, done, seed, pieces, errs = time.time(), 0, 70_000, 0, 0
while time.time() - t0 < args.seconds:
    seed += 1
    rng = np.random.default_rng(seed)
    k = seed % 6
    if k == 0:
        data = rand_fasta(rng, n_records=int(rng.integers(1, 400)), max_line=int(rng.integers(5, 300)), crlf=bool(rng.random() < 0.3),
                          tail_newline=bool(rng.random() < 0.7))
    elif k == 1:
        data = rand_fasta(rng, n_records=int(rng.integers(1, 200)), dirty=0.03, lead_blank=int(rng.integers(0, 3)))
    elif k == 2:
        data = rand_soup(rng, int(rng.integers(1, 3000)))
    elif k == 3:
        data = rand_soup(rng, int(rng.integers(1, 100000)), weights=[0.05, 3, 3, 1, 6, 6, 1, 0.002, 0.002])
    elif k == 4:   # long lines around the capacity, header lines among them
        parts = []
        for _ in range(int(rng.integers(2, 8))):
            ln = int(rng.choice([10, 100, CAP - 2, CAP - 1, CAP, CAP + 7]))
            parts.append((b">" if rng.random() < 0.4 else b"") + b"A" * ln + b"\n")
        data = b">s\nAC\n" + b"".join(parts)
    else:      # space runs over tile edges, headers behind them
        data = rand_soup(rng, int(rng.integers(1, 70000)), weights=[0.03, 0.05, 8, 2, 1, 1, 0.5, 0.01, 0.01])
    a = np.frombuffer(data, dtype=np.uint8)
    ca = bool(rng.random() < 0.5)
    fa = fas[ca]
    whole = FO.flat_parse(a, check_ascii=ca, line_cap=CAP)
    P = int(rng.integers(2, 9))
    cuts = sorted(int(x) for x in rng.integers(0, a.size + 1, P - 1))

    def summarize(piece):
        s = fa.shard_scan(upload(piece, 48), piece.size)
        got = (int(s.n_bytes), int(s.first_header), int(s.lead_kind), int(s.last_byte), int(s.tail_open))
        assert got == summary_of(piece, walk_cap=CAP), (got, summary_of(piece, walk_cap=CAP))
        return got

    def parse_region(region, pos_base, record_base=0, line_base=0):
        off = 16 + int(pos_base % 61)   # any alignment
        return DevFlat(fa, fa.parse(upload(np.ascontiguousarray(region), off), region.size, True, pos_base, line_base, record_base))

    try:
        recs, status, msg = stitch(a, cuts, ca, CAP, summarize=summarize, parse_region=parse_region)
        ok = recs == whole.records() and status == whole.status and (whole.status == 6 or msg == whole.message)
        why = "" if ok else f"{len(recs)} vs {whole.n_records}, status {status} vs {whole.status}, {msg!r} vs {whole.message!r}"
    except Exception as e:   # noqa: BLE001
        ok, why = False, repr(e)[:500]
    if not ok:
        print(f"MISMATCH seed={seed} kind={k} n={a.size} cuts={cuts} check_ascii={ca}: {why}")
        sys.exit(1)
    done += 1; pieces += P; errs += whole.status != 6
print(f"fasta shard campaign: {done} streams x 2..8 byte ranges ({pieces} ranges, {errs} streams end in an error) identical to the "
      f"sequential parse in {time.time()-t0:.0f} s")
