"""Regenerates tests/golden/*.json from the CPU oracle (the reference itself cannot run here:
Mojo is absent, see oracle/bzq_oracle.h).  The corpus *.fastq files under golden/corpus/ are the
reference's own test data (tests/test_data/fastq_parser, BioJava/Biopython suite).

Run from the repo root:  python tests/golden/make_golden.py
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

# schema each valid corpus file is parsed with by the reference's correctness test
# (tests/fastq/test_fastq_parser_correctness.mojo:142-444)
def schema_for(name):
    if "solexa" in name and "as_" not in name or name.endswith("as_solexa.fastq"):
        return "solexa"
    if name.endswith("as_illumina.fastq") or "illumina" in name and "as_" not in name:
        return "illumina_1.3"
    return "sanger" if "sanger" in name else "generic"


def digest(f):
    h = hashlib.sha256()
    for a in (f.header_start, f.seq_start, f.sep_start, f.qual_start, f.record_end, f.id_start,
              f.id_len, f.ends, f.id_ends, f.seq_bytes, f.qual_bytes, f.id_bytes):
        h.update(a.tobytes())
    return h.hexdigest()


def entry(data, cfg):
    f = O.flat_parse(data, cfg)
    return {"n_records": int(f.n_records), "term_code": int(f.term_code),
            "term_msg": f.term_msg.decode("latin-1"), "term_record": int(f.term_record),
            "consumed": int(f.consumed), "bases": int(f.seq_bytes.size), "digest": digest(f)}


def main():
    out = {}
    kseq = os.path.join(ROOT, "oracle", "_ref", "kseq_runner")
    for path in sorted(glob.glob(os.path.join(HERE, "corpus", "*.fastq"))):
        name = os.path.basename(path)
        data = open(path, "rb").read()
        e = {"size": len(data)}
        e["default"] = entry(data, O.make_config())
        e["validated_generic"] = entry(data, O.make_config(check_ascii=True, check_quality=True))
        sc = schema_for(name)
        e["schema"] = sc
        e["validated_schema"] = entry(data, O.make_config(check_ascii=True, check_quality=True, quality_schema=sc))
        e["validated_schema_simd32"] = entry(data, O.make_config(check_ascii=True, check_quality=True, quality_schema=sc, simd_width=32))
        e["cap64"] = entry(data, O.make_config(buffer_capacity=64))
        e["cap64_growth"] = entry(data, O.make_config(buffer_capacity=64, buffer_growth_enabled=True, buffer_max_capacity=1 << 20))
        if os.path.exists(kseq):
            r = subprocess.run([kseq, path], capture_output=True, text=True)
            if r.returncode == 0:
                a, b = r.stdout.split()
                e["kseq"] = [int(a), int(b)]
        out[name] = e
    json.dump(out, open(os.path.join(HERE, "corpus_expected.json"), "w"), indent=1, sort_keys=True)

    syn = {}
    for key, args in {"illumina150_64": (64, 150, 150, 33, 73, "generic"),
                      "ref_test_20": (20, 5, 12, 2, 25, "generic"),
                      "longread_8": (8, 200, 19800, 5, 30, "sanger"),
                      "sanger_1000": (1000, 50, 150, 0, 40, "sanger")}.items():
        buf = O.generate_synthetic(*args)
        f = O.flat_parse(buf, O.make_config(check_ascii=True, check_quality=True, quality_schema=args[5]))
        syn[key] = {"args": list(args), "size": int(buf.size), "sha256": hashlib.sha256(buf.tobytes()).hexdigest(),
                    "first_bytes": buf[:80].tobytes().decode("latin-1"), "n_records": int(f.n_records),
                    "term_code": int(f.term_code), "digest": digest(f)}
    json.dump(syn, open(os.path.join(HERE, "synthetic_expected.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(out), "corpus entries,", len(syn), "synthetic entries")


if __name__ == "__main__":
    main()
