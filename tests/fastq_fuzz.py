"""Seeded generators of adversarial FASTQ-like byte streams shared by the oracle and GPU tests."""
import numpy as np

SPACES = [9, 11, 12, 13, 28, 29, 30, 32]


def rand_record(rng, max_len=40, dirty=0.0, crlf=False):
    L = int(rng.integers(0, max_len + 1))
    idlen = int(rng.integers(0, 12))
    rid = bytes(rng.integers(48, 123, idlen).astype(np.uint8))
    if rng.random() < 0.3:
        rid = bytes(rng.choice(SPACES, int(rng.integers(0, 3))).astype(np.uint8)) + rid + \
              bytes(rng.choice(SPACES, int(rng.integers(0, 3))).astype(np.uint8))
    if rng.random() < 0.1:
        rid = rid[:len(rid) // 2] + b" " + rid[len(rid) // 2:]
    seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L))
    qual = bytes(rng.integers(33, 127, L).astype(np.uint8))
    plus = b"+" if rng.random() < 0.8 else b"+" + rid
    at = b"@"
    r = rng.random()
    if r < dirty:
        kind = int(rng.integers(0, 8))
        if kind == 0:
            at = b"r"
        elif kind == 1:
            plus = b"-" + plus[1:]
        elif kind == 2:
            qual = qual + b"!"
        elif kind == 3 and L > 0:
            s = bytearray(seq); s[int(rng.integers(0, L))] = int(rng.integers(128, 256)); seq = bytes(s)
        elif kind == 4 and L > 0:
            q = bytearray(qual); q[int(rng.integers(0, L))] = int(rng.choice([0, 9, 31, 32, 127, 200])); qual = bytes(q)
        elif kind == 5:
            rid = rid + bytes([int(rng.integers(128, 256))])
        elif kind == 6:
            seq = seq[:-1] if L else seq + b"A"
        elif kind == 7:
            return b"\n"
    nl = b"\r\n" if crlf else b"\n"
    return at + rid + nl + seq + nl + plus + nl + qual + nl


def rand_stream(rng, n_records=30, max_len=40, dirty=0.05, tail=None, crlf=False):
    parts = [rand_record(rng, max_len, dirty, crlf) for _ in range(n_records)]
    data = b"".join(parts)
    if tail is None:
        tail = int(rng.integers(0, 8))
    if tail == 1:      # last record without trailing newline
        data = data[:-1] if data.endswith(b"\n") else data
    elif tail == 2:    # truncated in the middle of something
        cut = int(rng.integers(0, max(1, min(len(data), 60))))
        data = data[:len(data) - cut]
    elif tail == 3:    # trailing blank lines
        data += b"\n" * int(rng.integers(1, 4))
    elif tail == 4:    # QUAL phase with only blanks after the '+' line
        data += b"@x\nAC\n+\n" + b" \t\r"[: int(rng.integers(0, 4))]
    elif tail == 5:    # junk with 1-2 lines
        data += b"@junk\nACGT"[: int(rng.integers(1, 11))]
    return data
