"""BGZF writer for the tests (SAM spec 4.1): every block an independent gzip member with the BC subfield."""
import struct
import zlib


def bgzf_block(data: bytes, level: int = 6, strategy: int = zlib.Z_DEFAULT_STRATEGY, mem_level: int = 8) -> bytes:
    assert len(data) <= 65536
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    payload = co.compress(data) + co.flush()
    bsize = 18 + len(payload) + 8
    assert bsize <= 65536, "does not fit one block"
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + payload +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def bgzf_compress(data: bytes, block: int = 65280, level: int = 6, strategy: int = zlib.Z_DEFAULT_STRATEGY, eof_marker: bool = True) -> bytes:
    out = []
    for i in range(0, len(data), block):
        piece = data[i:i + block]
        try:
            out.append(bgzf_block(piece, level, strategy))
        except AssertionError:   # incompressible: halve
            h = len(piece) // 2
            out.append(bgzf_block(piece[:h], level, strategy)); out.append(bgzf_block(piece[h:], level, strategy))
    if eof_marker:
        out.append(bgzf_block(b"", level))
    return b"".join(out)
