"""FASTA side of the CPU oracle -- TEST INFRASTRUCTURE ONLY (same rule as oracle/oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Two restatements of the reference's FASTA parser that must agree on every input:

* ``StreamFastaParser`` -- line by line, the way the reference runs: a ``BufferedReader`` window over a ``Reader`` that
  hands out at most ``chunk_size`` bytes per call (blazeseq/io/buffered.mojo:115-327), ``LineIterator.next_line``
  (buffered.mojo:600-638, 766-779) and ``FastaParser.next_record`` / ``_read_header_line``
  (blazeseq/fasta/parser.mojo:122-203).  Pure Python, small inputs only.
* ``flat_parse`` -- ctypes binding of oracle/fasta_oracle.c, the whole-buffer formulation the HIP kernels implement.

Parity status: pinned by source + the reference's FASTA known-answer tests (tests/test_oracle_fasta_kats.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfasta_oracle.so")

OK, NO_HEADER, ASCII_INVALID, EOF, LINE_TOO_LONG, EMPTY_SEQUENCE, NEED_MORE = 0, 1, 4, 6, 8, 11, 12
DEFAULT_CAPACITY = 256 * 1024   # CONSTS.mojo:26
_SPACES = frozenset((9, 10, 11, 12, 13, 28, 29, 30, 32))   # utils.mojo:267-289


def strip_spaces(b: bytes) -> bytes:   # utils.mojo:221-242
    lo, hi = 0, len(b)
    while lo < hi and b[lo] in _SPACES:
        lo += 1
    while hi > lo and b[hi - 1] in _SPACES:
        hi -= 1
    return b[lo:hi]


class FastaError(Exception):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code
        self.message = message


class _EOF(Exception):
    pass


class _BufferedReader:   # buffered.mojo:115-327
    def __init__(self, data: bytes, capacity: int, chunk_size: int):
        self.data, self.src_pos, self.chunk = data, 0, chunk_size
        self.buf = bytearray(capacity)
        self.cap, self.head, self.end, self.eof, self.stream_pos = capacity, 0, 0, False, 0
        self.fill()

    def available(self):
        return self.end - self.head

    def stream_position(self):
        return self.stream_pos + self.head

    def compact_from(self, pos):   # buffered.mojo:239-260
        if pos == 0:
            return
        if pos >= self.end:
            self.stream_pos += self.end
            self.head = self.end = 0
            return
        self.stream_pos += pos
        rem = self.end - pos
        self.buf[0:rem] = self.buf[pos:self.end]
        self.head = 0 if self.head < pos else self.head - pos
        self.end = rem

    def fill(self):   # buffered.mojo:262-281: EOF only once a read returns 0
        if self.eof:
            return 0
        space = self.cap - self.end
        if space == 0:
            return 0
        amt = min(space, len(self.data) - self.src_pos, self.chunk)
        self.buf[self.end:self.end + amt] = self.data[self.src_pos:self.src_pos + amt]
        self.src_pos += amt
        self.end += amt
        if amt == 0:
            self.eof = True
        return amt


class _LineIterator:   # buffered.mojo:521-779
    def __init__(self, data: bytes, capacity: int, chunk_size: int):
        self.b = _BufferedReader(data, capacity, chunk_size)
        self.line_number = 0
        self.file_position = 0

    def has_more(self):
        return self.b.available() > 0 or not self.b.eof

    def next_line(self) -> bytes:
        b = self.b
        self.file_position = b.stream_position()
        while True:
            if b.available() == 0:
                if b.eof:
                    raise _EOF()
                b.compact_from(b.head)
            b.fill()
            if b.available() == 0:
                raise _EOF()
            view = bytes(b.buf[b.head:b.end])
            at = view.find(b"\n")
            if at >= 0:
                end = at - 1 if at > 0 and view[at - 1] == 13 else at
                b.head += min(at + 1, b.available())
                self.line_number += 1
                return view[:end]
            if b.eof:
                end = len(view) - 1 if view[-1] == 13 else len(view)
                b.head += len(view)
                self.line_number += 1
                return view[:end]
            if len(view) >= b.cap:
                raise FastaError(LINE_TOO_LONG, "Line exceeds buffer capacity of %d bytes" % b.cap)
            b.compact_from(b.head)


def _parse_error(msg, rec, line, pos):   # errors.mojo:178-192
    s = msg
    if rec > 0:
        s += "\n  Record number: %d" % rec
    if line > 0:
        s += "\n  Line number: %d" % line
    if pos > 0:
        s += "\n  File position: %d" % pos
    return s


class StreamFastaParser:   # fasta/parser.mojo:60-203
    def __init__(self, data: bytes, check_ascii: bool = False, capacity: int = DEFAULT_CAPACITY, chunk_size: int = 1 << 30):
        self.lines = _LineIterator(bytes(data), capacity, chunk_size)
        self.record_number = 0
        self.pending: List[bytes] = []
        self.check_ascii = check_ascii

    def has_more(self):
        return len(self.pending) > 0 or self.lines.has_more()

    def _read_header_line(self) -> bytes:
        if self.pending:
            return self.pending.pop()
        while True:
            line = self.lines.next_line()
            t = strip_spaces(line)
            if len(t) == 0:
                continue
            if t[0] != 62:
                raise FastaError(NO_HEADER, _parse_error("FASTA: sequence id line does not start with '>'", self.record_number,
                                                         self.lines.line_number, self.lines.file_position))
            return strip_spaces(t[1:])

    def next_record(self) -> Tuple[bytes, bytes]:
        if not self.has_more():
            raise _EOF()
        rid = self._read_header_line()
        seq = bytearray()
        seq_start_line = self.lines.line_number + 1
        while True:
            try:
                line = strip_spaces(self.lines.next_line())
            except _EOF:
                break
            if len(line) > 0 and line[0] == 62:
                self.pending.append(strip_spaces(line[1:]))
                break
            seq += line
        if len(seq) == 0:
            raise FastaError(EMPTY_SEQUENCE, _parse_error("FASTA record has empty sequence", self.record_number + 1, seq_start_line,
                                                          self.lines.file_position))
        if self.check_ascii and (any(c >= 0x80 for c in rid) or any(c >= 0x80 for c in seq)):
            msg = "Non ASCII letters found"   # ValidationError.write_to, errors.mojo:223-234
            if self.record_number > 0:
                msg += "\n  Record number: %d" % self.record_number
            raise FastaError(ASCII_INVALID, msg)
        self.record_number += 1
        return rid, bytes(seq)

    def all_records(self):
        """(records, terminal code, message): every record up to the clean end or the first error."""
        out = []
        while True:
            try:
                out.append(self.next_record())
            except _EOF:
                return out, EOF, ""
            except FastaError as e:
                return out, e.code, e.message


# ---- flat (C) --------------------------------------------------------------------------------------------------------

class FaFlat(C.Structure):
    _fields_ = [
        ("n_records", C.c_int64),
        ("seq_bytes", C.c_void_p), ("seq_bytes_len", C.c_int64),
        ("id_bytes", C.c_void_p), ("id_bytes_len", C.c_int64),
        ("seq_ends", C.c_void_p), ("id_ends", C.c_void_p), ("hdr_pos", C.c_void_p),
        ("status", C.c_int32), ("_pad", C.c_int32),
        ("err_record", C.c_int64),
        ("err_record_number", C.c_int64), ("err_line_number", C.c_int64), ("err_file_position", C.c_int64),
        ("consumed", C.c_int64), ("lines_consumed", C.c_int64), ("total_lines", C.c_int64),
        ("message", C.c_char * 512),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fasta_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libfasta_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.fa_flat_parse.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.POINTER(FaFlat)]
        L.fa_flat_parse.restype = C.c_int
        L.fa_flat_free.argtypes = [C.POINTER(FaFlat)]
        L.fa_generate_synthetic.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64]
        L.fa_generate_synthetic.restype = C.c_int64
        L.fa_compute_num_reads_for_size.argtypes = [C.c_int64] * 4
        L.fa_compute_num_reads_for_size.restype = C.c_int64
        L.fa_bench_run.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.fa_bench_run.restype = C.c_int64
        _lib = L
    return _lib


class Flat:
    """Owned copy of a flat parse: columns as numpy arrays + the terminal event."""

    def __init__(self, f: FaFlat):
        n = int(f.n_records)
        self.n_records = n

        def arr(p, ln, dt):
            if ln == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8 if dt == np.uint8 else C.c_int64)), shape=(ln,)).astype(dt, copy=True)
        self.seq_bytes = arr(f.seq_bytes, int(f.seq_bytes_len), np.uint8)
        self.id_bytes = arr(f.id_bytes, int(f.id_bytes_len), np.uint8)
        self.seq_ends = arr(f.seq_ends, n, np.int64)
        self.id_ends = arr(f.id_ends, n, np.int64)
        self.hdr_pos = arr(f.hdr_pos, n, np.int64)
        self.status = int(f.status)
        self.err_record = int(f.err_record)
        self.err_record_number, self.err_line_number, self.err_file_position = int(f.err_record_number), int(f.err_line_number), int(f.err_file_position)
        self.consumed, self.lines_consumed, self.total_lines = int(f.consumed), int(f.lines_consumed), int(f.total_lines)
        self.message = f.message.decode("latin-1")

    def records(self):
        out, s0, i0 = [], 0, 0
        for r in range(self.n_records):
            s1, i1 = int(self.seq_ends[r]), int(self.id_ends[r])
            out.append((self.id_bytes[i0:i1].tobytes(), self.seq_bytes[s0:s1].tobytes()))
            s0, i0 = s1, i1
        return out


def flat_parse(data, check_ascii: bool = False, line_cap: int = DEFAULT_CAPACITY, is_eof: bool = True,
               record_base: int = 0, line_base: int = 0, pos_base: int = 0) -> Flat:
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    f = FaFlat()
    rc = lib().fa_flat_parse(a.ctypes.data, a.size, int(check_ascii), line_cap, int(is_eof), record_base, line_base, pos_base, C.byref(f))
    if rc:
        raise MemoryError("fa_flat_parse")
    try:
        return Flat(f)
    finally:
        lib().fa_flat_free(C.byref(f))


def generate_synthetic(num_reads: int, min_len: int, max_len: int, line_width: int = 60, gc_bias: float = 0.5) -> np.ndarray:
    n = lib().fa_generate_synthetic(num_reads, min_len, max_len, line_width, gc_bias, None, 0)
    if n < 0:
        raise ValueError("generate_synthetic_fasta_buffer: invalid arguments")
    out = np.empty(n, dtype=np.uint8)
    w = lib().fa_generate_synthetic(num_reads, min_len, max_len, line_width, gc_bias, out.ctypes.data, n)
    assert w == n, (w, n)
    return out


def compute_num_reads_for_size(target: int, min_len: int, max_len: int, line_width: int = 60) -> int:
    return int(lib().fa_compute_num_reads_for_size(target, min_len, max_len, line_width))
