"""Random FASTA-like byte streams for the FASTA parity tests (shared by CPU and GPU tests)."""
import numpy as np


def rand_fasta(rng, n_records=20, max_line=70, max_lines=6, dirty=0.0, crlf=False, tail_newline=True, lead_blank=0):
    """Mostly well-formed multi-line FASTA; `dirty` is the per-line chance of something odd (blank line, padded line,
    header without sequence, '>' inside a line, non-ASCII byte, leading-space header)."""
    out = []
    eol = b"\r\n" if crlf else b"\n"
    alphabet = np.frombuffer(b"ACGTNacgt-*", dtype=np.uint8)
    for _ in range(lead_blank):
        out.append(rng.choice([b"", b"  ", b"\t", b"\r"]) + b"\n")
    for r in range(n_records):
        hdr = b">seq%d some description %d" % (r, int(rng.integers(0, 1000)))
        if rng.random() < dirty:
            hdr = rng.choice([b">", b">   padded  ", b" \t>lead%d" % r, b">\tx\t", b">a>b", b">" + bytes([0x80 + int(rng.integers(0, 100))]) + b"z"])
        out.append(hdr + eol)
        nl = int(rng.integers(1, max_lines + 1))
        if rng.random() < dirty * 0.5:
            nl = 0
        for _ in range(nl):
            ln = int(rng.integers(1, max_line + 1))
            line = alphabet[rng.integers(0, alphabet.size, size=ln)].tobytes()
            if rng.random() < dirty:
                k = int(rng.integers(0, 6))
                if k == 0:
                    line = b""
                elif k == 1:
                    line = b"  " + line + b" \t "
                elif k == 2:
                    line = line[: ln // 2] + b"  " + line[ln // 2:]
                elif k == 3:
                    line = line[: ln // 2] + bytes([0x80 + int(rng.integers(0, 100))]) + line[ln // 2:]
                elif k == 4:
                    line = b"   "
                else:
                    line = line[:1] + b">" + line[1:]
            out.append(line + eol)
    data = b"".join(out)
    if not tail_newline and data.endswith(eol):
        data = data[: -len(eol)]
    return data


def rand_soup(rng, n, weights=None):
    """Uniform soup over the bytes that matter: structure is entirely accidental."""
    sym = np.frombuffer(b">\n \tAC\r\x80\x1c", dtype=np.uint8)
    p = np.array(weights if weights is not None else [2, 4, 3, 1, 6, 6, 1, 0.3, 0.3], dtype=float)
    return sym[rng.choice(sym.size, size=n, p=p / p.sum())].tobytes()
