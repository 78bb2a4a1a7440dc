"""FASTA path on one MI355X: the reference's benchmark input (benchmark/fasta-parser/generate_synthetic_fasta.mojo:
200-3800 bp, line width 60) generated on the device, parsed in place; also short 2-line records and long single-line
records.  Prints one JSON line per workload."""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import blazeseq_amd as B

ctx = B.FastaContext(B.FastaParserConfig(bool(int(sys.argv[1])) if len(sys.argv) > 1 else False))
for name, (nrec, lo, hi, lw) in {"ref_bench_200_3800_w60": (1_500_000, 200, 3800, 60), "reads_150_w150": (18_000_000, 150, 150, 150),
                                 "long_200_19800_w100000": (300_000, 200, 19800, 100000)}.items():
    t = ctx.generate_synthetic_device(nrec, lo, hi, lw)
    n = t.numel()
    ms = []
    for it in range(8):
        res = ctx.parse(int(t.data_ptr()), n, True)
        ms.append(res.kernel_ms)
    assert int(res.status) == 6 and int(res.n_records) == nrec, (res.status, res.n_records)
    best = float(np.median(ms[2:]))
    alg = n + int(res.seq_bytes) + int(res.id_bytes) + 16 * nrec
    print(json.dumps({"workload": name, "bytes": n, "records": nrec, "kernel_ms": round(best, 3), "GB_per_s": round(n / best / 1e6, 1),
                      "algorithmic_GB_per_s": round(alg / best / 1e6, 1), "frac_of_8TBps": round(alg / best / 1e6 / 8000, 3),
                      "check_ascii": ctx.config.check_ascii}))
    del t
    torch.cuda.empty_cache()
