"""Helpers of the gzip tests: gzip files built with zlib in every shape the decoder must take (members, levels, strategies,
header fields), and the device decode driven piece by piece."""
import ctypes as C
import struct
import zlib

import numpy as np

from blazeseq_amd import _lib as L


def gzip_member(data: bytes, level: int = 6, strategy: int = zlib.Z_DEFAULT_STRATEGY, name: bytes = b"", comment: bytes = b"", extra: bytes = b"",
                hcrc: bool = False, mem_level: int = 8, flush_every: int = 0) -> bytes:
    """One RFC 1952 member around a raw DEFLATE stream.  flush_every > 0: Z_FULL_FLUSH every so many input bytes (empty stored
    blocks and reset windows in the stream, like pigz -i)."""
    flg = (4 if extra else 0) | (8 if name else 0) | (16 if comment else 0) | (2 if hcrc else 0)
    hdr = bytes([0x1f, 0x8b, 8, flg, 0, 0, 0, 0, 0, 3])
    if extra:
        hdr += struct.pack("<H", len(extra)) + extra
    if name:
        hdr += name + b"\0"
    if comment:
        hdr += comment + b"\0"
    if hcrc:
        hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    if flush_every:
        body = b"".join(co.compress(data[i:i + flush_every]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), flush_every)) + co.flush()
    else:
        body = co.compress(data) + co.flush()
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


class DeviceGunzip:
    """A GzipDecoder with a device output buffer; decode(comp, piece) feeds the stream in pieces of `piece` bytes."""

    def __init__(self, ctx, out_capacity: int, chunk_bytes: int = 0):
        from blazeseq_amd.parser import GzipDecoder
        self.ctx, self.cap = ctx, out_capacity
        self.d_out = C.c_void_p()
        assert L.lib().bzq_device_alloc(ctx.h, out_capacity + 64, C.byref(self.d_out)) == 0
        self.dec = GzipDecoder(ctx, chunk_bytes)

    def decode(self, comp: bytes, piece: int = 0, ahead: int = 0) -> bytes:
        """ahead > 0: that many pieces are kept staged (bzq_gzip_stage) in front of the one being fed."""
        a = np.frombuffer(comp, dtype=np.uint8)
        piece = piece or max(1, a.size)
        out = []
        off = 0
        staged_to = 0   # pieces [0, staged_to) have been staged
        while True:
            k = off // piece
            while ahead and staged_to < k + 1 + ahead and staged_to * piece < a.size:
                if staged_to >= k:
                    self.dec.stage(a[staged_to * piece:(staged_to + 1) * piece])
                staged_to += 1
            part = a[off:off + piece]
            off += part.size
            last = off >= a.size
            while True:
                nb, more = self.dec.feed(part, last, self.d_out.value, self.cap)
                if nb:
                    h = np.empty(nb, dtype=np.uint8)
                    assert L.lib().bzq_copy_to_host(self.ctx.h, h.ctypes.data, self.d_out, nb) == 0
                    out.append(h.tobytes())
                if not more:
                    break
                part = a[:0]
            if last:
                break
        return b"".join(out)

    def close(self):
        self.dec.close()
        L.lib().bzq_device_free(self.ctx.h, self.d_out)


def sequencer_like(n_bytes: int, seed: int = 3) -> bytes:
    """FASTQ with what a sequencer's files have and the synthetic generator's do not: binned qualities in long runs (a match that
    overlaps itself: distance 1, length up to 258), reads sampled from a small genome (sources tens of KiB back), exact duplicates
    (matches as long as a read), poly-G tails."""
    rng = np.random.default_rng(seed)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 2_000_000)
    qv = np.frombuffer(b"F:,#", dtype=np.uint8)
    out, n, recent = [], 0, []
    i = 0
    while n < n_bytes:
        if recent and rng.random() < 0.15:
            seq = recent[int(rng.integers(len(recent)))]
        else:
            at = int(rng.integers(0, genome.size - 150))
            seq = genome[at:at + 150].copy()
            if rng.random() < 0.1:
                seq[150 - int(rng.integers(20, 90)):] = ord("G")
            miss = rng.random(150) < 0.003
            seq[miss] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(miss.sum()))
        recent.append(seq)
        if len(recent) > 64:
            recent.pop(0)
        runs = rng.geometric(0.07, 40)
        q = np.resize(np.repeat(rng.choice(qv, 40, p=[0.8, 0.12, 0.06, 0.02]), runs), 150)
        rec = b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT\n" % (1101 + i // 100000, 1000 + (i * 7) % 30000, 1000 + (i * 13) % 30000) + seq.tobytes() + b"\n+\n" + q.tobytes() + b"\n"
        out.append(rec); n += len(rec); i += 1
    return b"".join(out), i
