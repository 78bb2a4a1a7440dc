"""FASTA fuzz campaign (run by hand on a GPU box; not collected by pytest): random streams through bzq_fasta_parse vs
the flat oracle, whole-buffer and chunk mode, all switches.  `python tests/fuzz_campaign_fasta.py [seconds] [seed]`."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np

import blazeseq_amd as B
from fasta_fuzz import rand_fasta, rand_soup
from test_gpu_fasta import check_chunk

TILE = 16384


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctxs = {(a, c): B.FastaContext(B.FastaParserConfig(a, c)) for a in (False, True) for c in (32768, 256 * 1024)}
    t0, n, nbytes = time.time(), 0, 0
    kinds = {}
    while time.time() - t0 < budget:
        ctx = ctxs[(bool(rng.integers(0, 2)), int(rng.choice([32768, 256 * 1024])))]
        kind = int(rng.integers(0, 6))
        if kind == 0:
            data = rand_fasta(rng, int(rng.integers(1, 600)), int(rng.choice([5, 70, 300, 3000])), int(rng.integers(1, 8)),
                              dirty=float(rng.choice([0, 0.01, 0.1, 0.5])), crlf=bool(rng.integers(0, 2)), tail_newline=bool(rng.integers(0, 2)),
                              lead_blank=int(rng.integers(0, 3)))
        elif kind == 1:
            w = rng.random(9) ** 3
            w[1] *= float(rng.choice([1, 0.1, 0.01, 0.001]))   # newline density: down to one per ~100 kB
            data = rand_soup(rng, int(rng.choice([10, 1000, TILE, 3 * TILE + 5, 200_000])), w + 1e-6)
        elif kind == 2:   # long single-line records, some beyond the line capacity
            recs = []
            for i in range(int(rng.integers(1, 12))):
                L = int(rng.choice([1, 100, 16384, 32767, 32768, 40000, 262143, 262144, 300000]))
                pad = b" " * int(rng.integers(0, 3))
                recs.append(b">r%d\n" % i + pad + (b"ACGT" * (L // 4 + 1))[:max(L - len(pad), 1)] + rng.choice([b"\n", b"\r\n", b" \n"]))
            data = b"".join(recs)
            if rng.random() < 0.3:
                data = data.rstrip(b"\r\n ")
        elif kind == 3:   # runs of spaces / blank lines / headers around tile edges
            parts = []
            for _ in range(int(rng.integers(1, 10))):
                parts.append(rng.choice([b">", b" >", b">x y", b"AC", b"\n", b"\r\n", b" ", b"\t"]) * int(rng.choice([1, 2, 50, 5000, TILE - 1, TILE, TILE + 1])))
                parts.append(rng.choice([b"\n", b"", b"G", b">id\n"]))
            data = b"".join(parts)
        elif kind == 4:   # many tiny records (dense headers)
            data = b"".join(rng.choice([b">a\nA\n", b">\nC\n", b">b c\nG\nT\n", b">\n", b"\n"], p=[0.4, 0.3, 0.28, 0.01, 0.01]) for _ in range(int(rng.integers(1, 20000))))
        else:   # all byte values
            data = rng.integers(0, 256, size=int(rng.integers(1, 100_000)), dtype=np.uint8).tobytes()
            if rng.random() < 0.5:
                data = b">h\n" + data.replace(b">", b"<")
        try:
            check_chunk(ctx, data, True)
            cut = int(rng.integers(0, len(data) + 1))
            check_chunk(ctx, data[:cut], False, bases=(int(rng.integers(0, 1 << 40)), int(rng.integers(0, 1000)), int(rng.integers(0, 1000))))
            if n % 8 == 0 and len(data) < 300_000:   # the streaming parser over the same bytes, random chunk size
                from oracle import fasta as F
                want = F.flat_parse(data, ctx.config.check_ascii, ctx.config.line_capacity, True)
                p = B.FastaParser(data, ctx.config, chunk_bytes=int(rng.choice([64, 1000, 20_000, 1 << 20])))
                got, err = [], None
                try:
                    while True:
                        r = p.next_record()
                        got.append((r.id, r.sequence))
                except B.ParseError as e:
                    err = e
                p.close()
                assert got == want.records(), (len(got), want.n_records)
                if want.status == F.EOF:
                    assert err.code == F.EOF
                else:
                    assert err.code == want.status and err.message.decode("latin-1") == want.message, (err.message, want.message)
        except AssertionError:
            path = "gpurun_out/fasta_fuzz_fail_%d.bin" % n
            import os
            os.makedirs("gpurun_out", exist_ok=True)
            open(path, "wb").write(data)
            print("FAIL kind", kind, "len", len(data), "cfg", ctx.config, "saved", path)
            raise
        n += 1
        nbytes += len(data)
        kinds[kind] = kinds.get(kind, 0) + 1
    print(f"fasta campaign: {n} streams ({nbytes / 1e6:.0f} MB) x 2 modes identical, by kind {dict(sorted(kinds.items()))}, seed {seed}")


if __name__ == "__main__":
    main()
