"""bzq_bgzf_scan (C ABI, pure host function, no GPU): the block table of a BGZF buffer -- offsets, sizes, ISIZE, CRC-32 as the
trailers hold them; whole blocks only; the limits on output bytes and table entries; what is not a BGZF header."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from blazeseq_amd import _lib as L
from tests.bgzf_util import bgzf_block


def scan(buf: bytes, max_out=1 << 62, cap=None):
    a = np.frombuffer(buf, dtype=np.uint8) if buf else np.zeros(0, np.uint8)
    cap = cap if cap is not None else len(buf) // 28 + 1
    blocks = (L.BzqBgzfBlock * max(cap, 1))()
    n, consumed, out_bytes = C.c_int64(), C.c_uint64(), C.c_uint64()
    rc = L.lib().bzq_bgzf_scan(a.ctypes.data if a.size else None, a.size, max_out, blocks, cap, C.byref(n), C.byref(consumed), C.byref(out_bytes))
    return rc, [blocks[i] for i in range(n.value)], consumed.value, out_bytes.value


def test_table_matches_the_blocks():
    pieces = [b"ACGT" * 100, b"", b"x", bytes(range(256)) * 200, b"N" * 65536]
    parts = [bgzf_block(p, lvl) for p, lvl in zip(pieces, (6, 6, 0, 1, 9))]
    buf = b"".join(parts)
    rc, blocks, consumed, out_bytes = scan(buf)
    assert rc == 0 and len(blocks) == len(pieces) and consumed == len(buf) and out_bytes == sum(map(len, pieces))
    off = uoff = 0
    for b, p, c in zip(blocks, pieces, parts):
        assert (b.comp_offset, b.comp_size, b.out_size, b.out_offset) == (off, len(c), len(p), uoff)
        assert b.crc32 == zlib.crc32(p) & 0xFFFFFFFF
        off += len(c); uoff += len(p)


def test_whole_blocks_only_and_limits():
    parts = [bgzf_block(bytes([65 + i]) * (1000 * (i + 1))) for i in range(6)]
    buf = b"".join(parts)
    for cut in (0, 10, 27, len(parts[0]) - 1, len(parts[0]), len(parts[0]) + 5, len(buf) - 1):
        rc, blocks, consumed, _ = scan(buf[:cut])
        whole = 0
        while whole < len(parts) and sum(len(p) for p in parts[:whole + 1]) <= cut:
            whole += 1
        assert rc == 0 and len(blocks) == whole and consumed == sum(len(p) for p in parts[:whole])
    rc, blocks, consumed, out_bytes = scan(buf, max_out=1000 + 2000 + 2999)   # the third block (3000 bytes) does not fit
    assert rc == 0 and len(blocks) == 2 and out_bytes == 3000
    rc, blocks, consumed, _ = scan(buf, cap=4)
    assert rc == 0 and len(blocks) == 4 and consumed == sum(len(p) for p in parts[:4])


def test_not_a_bgzf_header():
    good = bgzf_block(b"hello world")
    import gzip
    for bad in (gzip.compress(b"hello world" * 10), b"\x1f\x8b" + b"\0" * 40, b"plain text, no gzip magic at all ......."):
        rc, blocks, consumed, _ = scan(bad)
        assert rc < 0 and not blocks and consumed == 0
    # a good block, then garbage: the table ends at the good block and the error names the offset behind it
    rc, blocks, consumed, _ = scan(good + b"garbage-garbage-garbage-garbage-garbage")
    assert rc < 0 and len(blocks) == 1 and consumed == len(good)
    # a header that claims more than 64 KiB of output
    big = bytearray(good)
    big[-4:] = struct.pack("<I", 65537)
    rc, blocks, _, _ = scan(bytes(big))
    assert rc < 0 and not blocks
