"""Regenerates tests/golden/fasta_expected.json from the CPU oracle (oracle/fasta_oracle.c; the reference itself cannot run
here).  The files under golden/fasta/ are the reference's own FASTA test data (Biopython Tests/Fasta); a handful of
constructed streams pin the error paths and the generator.

Run from the repo root:  python tests/golden/make_golden_fasta.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import fasta as F  # noqa: E402

CONSTRUCTED = {
    "crlf_blank_lines": b"\r\n\r\n>a  b \r\nAC GT\r\n\r\n  \r\n>c\r\n T \r\n",
    "empty_second": b">id1\nACGT\n>id2\n>id3\nGGGG\n",
    "no_header": b"\n  \nACGT\n>id1\nACGT\n",
    "non_ascii_third": b">a\nAC\n>b\nGT\n>c\nA\x80C\n>d\nTT\n",
    "gt_inside": b">a>b\nAC>GT\n  >c  d \n A C \n",
    "unterminated": b">id1\nACG\nTTA",
}


def entry(data: bytes, check_ascii: bool):
    f = F.flat_parse(data, check_ascii)
    h = hashlib.sha256()
    for a in (f.id_bytes, f.id_ends, f.seq_bytes, f.seq_ends, f.hdr_pos):
        h.update(a.tobytes())
    return {"n_records": f.n_records, "status": f.status, "message": f.message, "seq_bytes": int(f.seq_bytes.size),
            "id_bytes": int(f.id_bytes.size), "digest": h.hexdigest()}


def main():
    out = {}
    for name in sorted(os.listdir(os.path.join(HERE, "fasta"))):
        if name.endswith(".md"):
            continue
        data = open(os.path.join(HERE, "fasta", name), "rb").read()
        out["file:" + name] = {"plain": entry(data, False), "check_ascii": entry(data, True)}
    for name, data in CONSTRUCTED.items():
        out["stream:" + name] = {"plain": entry(data, False), "check_ascii": entry(data, True)}
    gen = F.generate_synthetic(2000, 5, 400, 60).tobytes()
    out["generator:2000x5-400w60"] = {"sha256": hashlib.sha256(gen).hexdigest(), "bytes": len(gen), "plain": entry(gen, False)}
    with open(os.path.join(HERE, "fasta_expected.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", len(out), "entries")


if __name__ == "__main__":
    main()
