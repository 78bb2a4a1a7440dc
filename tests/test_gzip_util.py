"""CPU: the gzip files the GPU tests feed the device decoder are what they claim to be -- members built by tests/gzip_util.py
decode with zlib / the gzip module to the bytes they were made from, in every layout the tests use (header fields, full
flushes, levels, strategies, memLevel).  The device tests then compare the device against those bytes."""
import gzip
import zlib

import numpy as np

from tests.gzip_util import gzip_member


def test_members_are_valid_gzip_in_every_layout():
    rng = np.random.default_rng(1)
    payloads = [b"", b"A", bytes(rng.integers(0, 256, 70000, dtype=np.uint8)), b"ACGT" * 50000, bytes(rng.integers(65, 70, 200000, dtype=np.uint8))]
    blob, want = b"", b""
    for i, data in enumerate(payloads):
        for level, strategy in [(6, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (9, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY)]:
            m = gzip_member(data, level, strategy, name=b"n%d" % i if i % 2 else b"", comment=b"c" if i == 2 else b"", extra=b"XY\x02\x00ab" if i == 3 else b"",
                            hcrc=(i == 4), mem_level=1 + (i * 2) % 9, flush_every=30000 if i >= 2 else 0)
            assert gzip.decompress(m) == data
            blob += m; want += data
    assert gzip.decompress(blob) == want   # concatenated members are one gzip file (RFC 1952 2.2)


def test_flushed_members_hold_stored_empty_blocks_and_reset_windows():
    """flush_every inserts Z_FULL_FLUSH points: an empty stored block (00 00 ff ff) at a byte edge, the pattern pigz -i writes."""
    data = b"ACGTTGCA" * 20000
    m = gzip_member(data, 6, flush_every=40000)
    assert m.count(b"\x00\x00\xff\xff") >= 3 and gzip.decompress(m) == data
