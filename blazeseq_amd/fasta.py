"""FASTA side of blazeseq_amd: host mirror of the reference's ``FastaParser`` / ``FastaRecord`` / fasta
``ParserConfig`` (blazeseq/fasta/parser.mojo:24-244, blazeseq/fasta/record.mojo:11-144, blazeseq/fasta/definition.mojo)
over the ``bzq_fasta_*`` entry points of libblazeseq_hip.so.  Every byte is classified, stripped and packed by the HIP
kernels in csrc/bzq_fasta.hpp; this module only feeds chunks and slices the columns that come back.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import io
import os
from dataclasses import dataclass
from typing import Iterator, Optional, Tuple

import numpy as np

from . import _lib as L
from .parser import ParseError, EOFError_

_SPACES = bytes((9, 10, 11, 12, 13, 28, 29, 30, 32))   # is_posix_space, utils.mojo:267-289
DEFAULT_CHUNK_BYTES = 64 << 20


@dataclass(frozen=True)
class FastaParserConfig:
    """fasta ``ParserConfig`` (fasta/parser.mojo:24-35) + the LineIterator capacity it implies (CONSTS.mojo:26)."""
    check_ascii: bool = False
    line_capacity: int = 256 * 1024


@dataclass(frozen=True)
class Definition:   # fasta/definition.mojo:4-18
    Id: bytes
    Description: Optional[bytes]


class FastaRecord:
    """Owned FASTA record (fasta/record.mojo:11-144): id without '>', sequence on one logical line."""
    __slots__ = ("_id", "_sequence")

    def __init__(self, id, sequence):
        self._id = id.encode() if isinstance(id, str) else bytes(id)
        self._sequence = sequence.encode() if isinstance(sequence, str) else bytes(sequence)

    @property
    def id(self) -> bytes:
        return self._id

    @property
    def sequence(self) -> bytes:
        return self._sequence

    def definition(self) -> Definition:   # record.mojo:86-99: split on single spaces, the rest glued without them
        parts = self._id.split(b" ")
        if len(parts) > 1:
            return Definition(parts[0].strip(), b"".join(parts[1:]).strip(_SPACES))
        return Definition(parts[0].strip(), None)

    def byte_len(self) -> int:   # record.mojo:101-105
        return 1 + len(self._id) + 1 + len(self._sequence) + 1

    def write(self, line_width: int = 60) -> bytes:   # record.mojo:107-124
        w = line_width if line_width > 0 else len(self._sequence)
        out = [b">", self._id, b"\n"]
        for i in range(0, len(self._sequence), max(w, 1)):
            out += [self._sequence[i:i + w], b"\n"]
        return b"".join(out)

    def __len__(self):
        return len(self._sequence)

    def __eq__(self, other):   # record.mojo:138-139: equality is on the sequence only
        return isinstance(other, FastaRecord) and self._sequence == other._sequence

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._sequence)

    def __repr__(self):
        return self.write().decode("latin-1")


class FastaContext:
    """One ``bzq_fasta`` handle (device arenas + stream)."""

    def __init__(self, config: FastaParserConfig = FastaParserConfig(), device: int = 0):
        self._lib = L.lib()
        self._h = C.c_void_p()
        cfg = L.BzqFastaConfig(int(config.check_ascii), 0, int(config.line_capacity))
        rc = self._lib.bzq_fasta_create(device, C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(self._lib.bzq_fasta_last_error(None).decode())
        self.config = config

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.bzq_fasta_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError(self._lib.bzq_fasta_last_error(self._h).decode())

    def parse(self, data, n: int, is_eof: bool, stream_pos: int = 0, line_base: int = 0, record_base: int = 0) -> L.BzqFastaChunk:
        """data: host buffer (bytes / numpy / int address) or a device pointer (int)."""
        keep = None
        if isinstance(data, int):
            ptr = data
        else:
            keep = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
            ptr = keep.ctypes.data
        out = L.BzqFastaChunk()
        self._check(self._lib.bzq_fasta_parse(self._h, C.c_void_p(ptr), n, int(is_eof), stream_pos, line_base, record_base, C.byref(out)))
        del keep
        return out

    def shard_scan(self, d_ptr: int, n: int) -> L.BzqFastaShardSummary:
        """What one byte range tells the other ranks (bzq_fasta_shard_scan)."""
        out = L.BzqFastaShardSummary()
        self._check(self._lib.bzq_fasta_shard_scan(self._h, C.c_void_p(d_ptr), n, C.byref(out)))
        return out

    def shard_stitch(self, comm_ctx, d_ptr: int, n: int, capacity: int) -> L.BzqFastaShardResult:
        """bzq_fasta_shard_stitch: this rank's byte range of the stream -> its records + the global outcome.  comm_ctx: the
        parser.Context that holds the communicator (comm_init / comm_init_shm), or None for one rank."""
        out = L.BzqFastaShardResult()
        ch = comm_ctx.h if comm_ctx is not None else None
        self._check(self._lib.bzq_fasta_shard_stitch(ch, self._h, C.c_void_p(d_ptr), n, capacity, C.byref(out)))
        return out

    def error_text(self) -> bytes:
        n = self._lib.bzq_fasta_format_error(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self._lib.bzq_fasta_format_error(self._h, buf, n + 1)
        return buf.raw[:n]

    def to_host(self, d_ptr: int, count: int, dtype) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if count:
            self._check(self._lib.bzq_fasta_copy_to_host(self._h, out.ctypes.data, C.c_void_p(d_ptr), out.nbytes))
        return out

    def columns(self, res: L.BzqFastaChunk):
        """(id_bytes, id_ends, seq_bytes, seq_ends, hdr_pos) of a chunk result, on the host."""
        n = int(res.n_records)
        return (self.to_host(res.d_id_bytes, int(res.id_bytes), np.uint8), self.to_host(res.d_id_ends, n, np.int64),
                self.to_host(res.d_seq_bytes, int(res.seq_bytes), np.uint8), self.to_host(res.d_seq_ends, n, np.int64),
                self.to_host(res.d_hdr_pos, n, np.int64))

    def generate_synthetic_device(self, num_reads: int, min_len: int, max_len: int, line_width: int = 60, first: int = 0,
                                  count: Optional[int] = None):
        """generate_synthetic_fasta_buffer (utils.mojo:1033-1139) into a fresh torch uint8 CUDA tensor."""
        import torch
        count = num_reads - first if count is None else count
        nbytes = C.c_uint64()
        self._check(self._lib.bzq_fasta_generate_synthetic_device(self._h, num_reads, first, count, min_len, max_len, line_width, None, 0, C.byref(nbytes)))
        t = torch.empty(int(nbytes.value) + 64, dtype=torch.uint8, device="cuda")
        self._check(self._lib.bzq_fasta_generate_synthetic_device(self._h, num_reads, first, count, min_len, max_len, line_width,
                                                                  C.c_void_p(t.data_ptr()), t.numel(), C.byref(nbytes)))
        return t[: int(nbytes.value)]


class FastaIngest:
    """``bzq_fasta_ingest_*``: the native file -> pinned -> device pipeline (reader threads, gzip / BGZF inflated on the
    way) in front of the FASTA parser.  ``next()`` parses the next chunk of the file; every record it delivers is taken."""

    def __init__(self, ctx: FastaContext, path: str, chunk_bytes: int = 256 << 20, n_threads: int = 0):
        self._ctx = ctx
        self._g = C.c_void_p()
        rc = ctx._lib.bzq_fasta_ingest_open(ctx._h, os.fspath(path).encode(), int(chunk_bytes), int(n_threads), C.byref(self._g))
        if rc != 0:
            raise RuntimeError(ctx._lib.bzq_fasta_last_error(ctx._h).decode())
        self.stream_pos = 0

    def next(self) -> L.BzqFastaChunk:
        out, pos = L.BzqFastaChunk(), C.c_uint64()
        self._ctx._check(self._ctx._lib.bzq_fasta_ingest_next(self._g, C.byref(out), C.byref(pos)))
        self.stream_pos = int(pos.value)
        return out

    def stats(self) -> L.BzqIngestStats:
        st = L.BzqIngestStats()
        self._ctx._lib.bzq_fasta_ingest_get_stats(self._g, C.byref(st))
        return st

    def close(self):
        if getattr(self, "_g", None) is not None and self._g:
            self._ctx._lib.bzq_fasta_ingest_close(self._g)
            self._g = C.c_void_p()

    __del__ = close


class FastaParser:
    """``FastaParser[R, config]`` (fasta/parser.mojo:60-203): ``next_record()``, ``has_more()``, ``records()`` /
    iteration.  ``source``: bytes-like, a path, or a binary file object.  Records come out of the device columns a chunk
    at a time; a record longer than the chunk makes the chunk grow."""

    def __init__(self, source, config: FastaParserConfig = FastaParserConfig(), chunk_bytes: int = DEFAULT_CHUNK_BYTES, device: int = 0,
                 check_ascii: Optional[bool] = None, native_ingest: bool = True, reader_threads: int = 0):
        if check_ascii is not None:
            config = FastaParserConfig(check_ascii, config.line_capacity)
        self._ctx = FastaContext(config, device)
        self._own_file = False
        self._ingest: Optional[FastaIngest] = None
        self._fh = None
        if native_ingest and isinstance(source, (str, os.PathLike)) and os.path.isfile(source):
            # a file goes through the native pipeline (plain, gzip and BGZF alike)
            self._ingest = FastaIngest(self._ctx, source, max(int(chunk_bytes), 1 << 16), reader_threads)
        elif isinstance(source, (str, os.PathLike)):
            self._fh = open(source, "rb")
            self._own_file = True
            if self._fh.read(2) == b"\x1f\x8b":   # gzip / BGZF, like the reference's GZFile reader (io/readers.mojo:283-377)
                import gzip
                self._fh.close()
                self._fh = gzip.open(source, "rb")
            else:
                self._fh.seek(0)
        elif hasattr(source, "read"):
            self._fh = source
        else:
            self._fh = io.BytesIO(bytes(source))
        self._chunk = max(int(chunk_bytes), 1)
        self._carry = b""
        self._src_eof = False
        self._pos = 0          # stream offset of the carry
        self._lines = 0
        self._record_number = 0
        self._queue: list = []
        self._qi = 0
        self._done = False
        self._pending_error: Optional[ParseError] = None

    def close(self):
        if self._ingest is not None:
            self._ingest.close()
            self._ingest = None
        if self._own_file:
            self._fh.close()
            self._own_file = False
        self._ctx.close()

    def _take(self, res) -> int:
        """Records of a chunk result into the queue."""
        n = int(res.n_records)
        if n:
            idb, ide, sqb, sqe, _ = self._ctx.columns(res)
            i0 = s0 = 0
            recs = []
            for r in range(n):
                i1, s1 = int(ide[r]), int(sqe[r])
                recs.append(FastaRecord(idb[i0:i1].tobytes(), sqb[s0:s1].tobytes()))
                i0, s0 = i1, s1
            self._queue, self._qi = recs, 0
            self._record_number += n
        return n

    def _fill(self):
        """Parse the next chunk into the record queue (or set the terminal state)."""
        while self._ingest is not None and not self._done and self._qi >= len(self._queue):
            res = self._ingest.next()
            self._take(res)
            status = int(res.status)
            if status == L.EOF:
                self._done = True
            elif status != L.OK:
                self._done = True
                self._pending_error = ParseError(status, self._ctx.error_text())
        while self._ingest is None and not self._done and self._qi >= len(self._queue):
            want = self._chunk - len(self._carry)
            fresh = b"" if self._src_eof or want <= 0 else self._fh.read(want)
            if not self._src_eof and want > 0 and len(fresh) < want:
                # a short read is not EOF for every file object: ask once more
                more = self._fh.read(1) if fresh else b""
                if more:
                    fresh += more
                else:
                    self._src_eof = True
            data = self._carry + fresh
            res = self._ctx.parse(data, len(data), self._src_eof, self._pos, self._lines, self._record_number)
            status = int(res.status)
            self._take(res)
            if status in (L.OK, L.FASTA_NEED_MORE):
                used = int(res.bytes_consumed)
                self._carry = data[used:]
                self._pos += used
                self._lines += int(res.lines_consumed)
                if status == L.FASTA_NEED_MORE and len(self._carry) >= self._chunk:
                    self._chunk *= 2   # the open record does not fit yet
            elif status == L.EOF:
                self._done = True
            else:
                self._done = True
                self._pending_error = ParseError(status, self._ctx.error_text())

    def has_more(self) -> bool:   # parser.mojo:104-107
        self._fill()
        return self._qi < len(self._queue) or self._pending_error is not None

    def next_record(self) -> FastaRecord:   # parser.mojo:122-172
        self._fill()
        if self._qi < len(self._queue):
            rec = self._queue[self._qi]
            self._qi += 1
            return rec
        if self._pending_error is not None:
            e, self._pending_error = self._pending_error, None
            raise e
        raise EOFError_(L.EOF, b"EOF")

    def records(self) -> Iterator[FastaRecord]:
        """``for rec in parser`` (parser.mojo:174-175, 205-244): EOF ends the iteration, and so does a parse error --
        the reference's iterator prints its text and stops; ``next_record()`` is the call that raises it."""
        while True:
            try:
                yield self.next_record()
            except EOFError_:
                return
            except ParseError as e:
                print(e.message.decode("latin-1"))
                return

    __iter__ = records
