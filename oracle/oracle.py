"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``blazeseq_amd`` never does (tests/test_no_oracle_in_product.py
enforces that).  See oracle/bzq_oracle.h for what is restated and how parity is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbzq_oracle.so")

OK, ID_NO_AT, SEP_NO_PLUS, SEQ_QUAL_LEN_MISMATCH, ASCII_INVALID, QUALITY_OUT_OF_RANGE, EOF, \
    UNEXPECTED_EOF, BUFFER_EXCEEDED, BUFFER_AT_MAX, OTHER = range(11)


class OrcConfig(C.Structure):
    _fields_ = [
        ("buffer_capacity", C.c_int64),
        ("buffer_max_capacity", C.c_int64),
        ("buffer_growth_enabled", C.c_int32),
        ("check_ascii", C.c_int32),
        ("check_quality", C.c_int32),
        ("q_lower", C.c_uint8),
        ("q_upper", C.c_uint8),
        ("q_offset", C.c_uint8),
        ("_pad", C.c_uint8),
        ("simd_width", C.c_int32),
        ("batch_size", C.c_int32),
    ]


class OrcView(C.Structure):
    _fields_ = [
        ("id", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p),
        ("id_len", C.c_int64), ("seq_len", C.c_int64), ("qual_len", C.c_int64),
        ("rec_pos", C.c_int64), ("off", C.c_int64 * 5), ("id_pos", C.c_int64),
    ]


class OrcBatch(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("id_bytes", C.c_void_p), ("id_bytes_len", C.c_int64),
        ("qual_bytes", C.c_void_p), ("qual_bytes_len", C.c_int64),
        ("seq_bytes", C.c_void_p), ("seq_bytes_len", C.c_int64),
        ("id_ends", C.c_void_p), ("ends", C.c_void_p),
        ("quality_offset", C.c_uint8),
        ("cap_n", C.c_int64), ("cap_id", C.c_int64), ("cap_qual", C.c_int64), ("cap_seq", C.c_int64),
    ]


class OrcFlat(C.Structure):
    _fields_ = [
        ("n_records", C.c_int64),
        ("header_start", C.c_void_p), ("seq_start", C.c_void_p), ("sep_start", C.c_void_p),
        ("qual_start", C.c_void_p), ("record_end", C.c_void_p),
        ("id_start", C.c_void_p), ("id_len", C.c_void_p),
        ("seq_bytes", C.c_void_p), ("seq_bytes_len", C.c_int64),
        ("qual_bytes", C.c_void_p), ("qual_bytes_len", C.c_int64),
        ("id_bytes", C.c_void_p), ("id_bytes_len", C.c_int64),
        ("ends", C.c_void_p), ("id_ends", C.c_void_p),
        ("term_code", C.c_int32), ("_pad", C.c_int32),
        ("term_record", C.c_int64), ("consumed", C.c_int64), ("n_newlines", C.c_int64),
        ("term_msg", C.c_char * 1400),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle (and, if /root/reference is present, oracle/_ref/kseq_runner)."""
    src = os.path.join(_HERE, "bzq_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libbzq_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(_HERE, "_ref", "kseq_runner")):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_config_default.argtypes = [C.POINTER(OrcConfig)]
        L.orc_schema_from_name.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.orc_schema_from_name.restype = C.c_int
        L.orc_message_for_code.argtypes = [C.c_int]
        L.orc_message_for_code.restype = C.c_char_p
        L.orc_parser_new.argtypes = [C.c_void_p, C.c_int64, C.POINTER(OrcConfig)]
        L.orc_parser_new.restype = C.c_void_p
        L.orc_parser_free.argtypes = [C.c_void_p]
        L.orc_parser_has_more.argtypes = [C.c_void_p]
        L.orc_parser_has_more.restype = C.c_int
        L.orc_parser_next_view.argtypes = [C.c_void_p, C.POINTER(OrcView)]
        L.orc_parser_next_view.restype = C.c_int
        L.orc_parser_next_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(OrcBatch)]
        L.orc_parser_next_batch.restype = C.c_int
        L.orc_parser_error.argtypes = [C.c_void_p]
        L.orc_parser_error.restype = C.c_char_p
        L.orc_parser_line_number.argtypes = [C.c_void_p]
        L.orc_parser_line_number.restype = C.c_int64
        L.orc_parser_stream_position.argtypes = [C.c_void_p]
        L.orc_parser_stream_position.restype = C.c_int64
        L.orc_parser_capacity.argtypes = [C.c_void_p]
        L.orc_parser_capacity.restype = C.c_int64
        L.orc_batch_init.argtypes = [C.POINTER(OrcBatch)]
        L.orc_batch_free.argtypes = [C.POINTER(OrcBatch)]
        L.orc_flat_parse.argtypes = [C.c_void_p, C.c_int64, C.POINTER(OrcConfig), C.c_int, C.POINTER(OrcFlat)]
        L.orc_flat_parse.restype = C.c_int
        L.orc_flat_free.argtypes = [C.POINTER(OrcFlat)]
        L.orc_generate_synthetic.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                             C.c_int64, C.c_int64, C.c_char_p, C.c_double,
                                             C.c_void_p, C.c_int64]
        L.orc_generate_synthetic.restype = C.c_int64
        L.orc_compute_num_reads_for_size.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        L.orc_compute_num_reads_for_size.restype = C.c_int64
        L.orc_bench_run.argtypes = [C.c_void_p, C.c_int64, C.POINTER(OrcConfig), C.c_int, C.POINTER(C.c_int64)]
        L.orc_bench_run.restype = C.c_int64
        L.orc_nw_score.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        L.orc_nw_score.restype = C.c_int32
        L.orc_pipeline_run.argtypes = [C.c_void_p, C.c_int64, C.POINTER(OrcConfig), C.c_char_p, C.c_int64, C.c_int64, C.c_void_p,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_pipeline_run.restype = C.c_int64
        _lib = L
    return _lib


def schema(name: str):
    lo, up, off = C.c_uint8(), C.c_uint8(), C.c_uint8()
    known = lib().orc_schema_from_name(name.encode(), C.byref(lo), C.byref(up), C.byref(off))
    return lo.value, up.value, off.value, bool(known)


def make_config(buffer_capacity: int = 256 * 1024, buffer_max_capacity: int = 1 << 30,
                buffer_growth_enabled: bool = False, check_ascii: bool = False,
                check_quality: bool = False, quality_schema: str = "generic", simd_width: int = 0,
                batch_size: int = 4096) -> OrcConfig:
    c = OrcConfig()
    lib().orc_config_default(C.byref(c))
    c.buffer_capacity = buffer_capacity
    c.buffer_max_capacity = buffer_max_capacity
    c.buffer_growth_enabled = int(buffer_growth_enabled)
    c.check_ascii = int(check_ascii)
    c.check_quality = int(check_quality)
    c.q_lower, c.q_upper, c.q_offset, _ = schema(quality_schema)
    c.simd_width = simd_width
    c.batch_size = batch_size
    return c


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    if isinstance(data, str):
        data = data.encode("latin-1")
    return np.frombuffer(bytes(data), dtype=np.uint8)


@dataclass
class View:
    id: bytes
    seq: bytes
    qual: bytes
    rec_pos: int
    off: tuple
    id_pos: int


@dataclass
class Batch:
    n: int
    id_bytes: bytes
    qual_bytes: bytes
    seq_bytes: bytes
    id_ends: List[int]
    ends: List[int]
    quality_offset: int = 33

    def __len__(self):
        return self.n

    def get_record(self, i: int):
        """FastqBatch.get_record, fastq/record_batch.mojo:116-150."""
        if i < 0 or i >= self.n:
            raise IndexError("FastqBatch.get_record index out of range")
        i0 = 0 if i == 0 else self.id_ends[i - 1]
        s0 = 0 if i == 0 else self.ends[i - 1]
        return (self.id_bytes[i0:self.id_ends[i]], self.seq_bytes[s0:self.ends[i]],
                self.qual_bytes[s0:self.ends[i]])


class OracleError(Exception):
    def __init__(self, code: int, message: bytes):
        super().__init__(message.decode("latin-1"))
        self.code = code
        self.message = message


class StreamParser:
    """FastqParser[MemoryReader, config] of the reference, restated (streaming)."""

    def __init__(self, data, config: Optional[OrcConfig] = None):
        self._data = _as_u8(data)
        self.config = config if config is not None else make_config()
        self._p = lib().orc_parser_new(self._data.ctypes.data, self._data.size, C.byref(self.config))

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_parser_free(self._p)
            self._p = None

    def has_more(self) -> bool:
        return bool(lib().orc_parser_has_more(self._p))

    def next_view(self) -> View:
        v = OrcView()
        rc = lib().orc_parser_next_view(self._p, C.byref(v))
        if rc != OK:
            raise OracleError(rc, lib().orc_parser_error(self._p))
        return View(C.string_at(v.id, v.id_len), C.string_at(v.seq, v.seq_len),
                    C.string_at(v.qual, v.qual_len), v.rec_pos, tuple(v.off), v.id_pos)

    def next_batch(self, max_records: int = 4096) -> Batch:
        b = OrcBatch()
        lib().orc_batch_init(C.byref(b))
        try:
            rc = lib().orc_parser_next_batch(self._p, max_records, C.byref(b))
            if rc != OK:
                raise OracleError(rc, lib().orc_parser_error(self._p))
            n = b.n
            id_ends = list(np.ctypeslib.as_array(C.cast(b.id_ends, C.POINTER(C.c_int64)), (n,))) if n else []
            ends = list(np.ctypeslib.as_array(C.cast(b.ends, C.POINTER(C.c_int64)), (n,))) if n else []
            return Batch(n, C.string_at(b.id_bytes, b.id_bytes_len) if n else b"",
                         C.string_at(b.qual_bytes, b.qual_bytes_len) if n else b"",
                         C.string_at(b.seq_bytes, b.seq_bytes_len) if n else b"",
                         [int(x) for x in id_ends], [int(x) for x in ends], b.quality_offset)
        finally:
            lib().orc_batch_free(C.byref(b))

    def views(self):
        """_FastqParserViewIter, parser.mojo:628-661: any error ends the iteration."""
        while True:
            try:
                yield self.next_view()
            except OracleError:
                return

    def batches(self, max_records: Optional[int] = None):
        """_FastqParserBatchIter, parser.mojo:700-735."""
        limit = max_records if max_records else self.config.batch_size
        while self.has_more():
            try:
                b = self.next_batch(limit)
            except OracleError:
                return
            if len(b) == 0:
                return
            yield b

    @property
    def line_number(self):
        return lib().orc_parser_line_number(self._p)

    @property
    def capacity(self):
        return lib().orc_parser_capacity(self._p)

    def stream_all(self):
        """Run next_view to the terminal event: ([View...], term_code, term_message)."""
        out = []
        while True:
            try:
                out.append(self.next_view())
            except OracleError as e:
                return out, e.code, e.message


@dataclass
class Flat:
    n_records: int
    header_start: np.ndarray
    seq_start: np.ndarray
    sep_start: np.ndarray
    qual_start: np.ndarray
    record_end: np.ndarray
    id_start: np.ndarray
    id_len: np.ndarray
    seq_bytes: np.ndarray
    qual_bytes: np.ndarray
    id_bytes: np.ndarray
    ends: np.ndarray
    id_ends: np.ndarray
    term_code: int
    term_record: int
    consumed: int
    n_newlines: int
    term_msg: bytes


def flat_parse(data, config: Optional[OrcConfig] = None, is_eof: bool = True) -> Flat:
    d = _as_u8(data)
    cfg = config if config is not None else make_config()
    f = OrcFlat()
    lib().orc_flat_parse(d.ctypes.data, d.size, C.byref(cfg), int(is_eof), C.byref(f))
    try:
        n = f.n_records

        def i64(p):
            if not n:
                return np.zeros(0, dtype=np.int64)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64)), (n,)).copy()

        def u8(p, ln):
            if not ln:
                return np.zeros(0, dtype=np.uint8)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (ln,)).copy()

        return Flat(n, i64(f.header_start), i64(f.seq_start), i64(f.sep_start), i64(f.qual_start),
                    i64(f.record_end), i64(f.id_start), i64(f.id_len),
                    u8(f.seq_bytes, f.seq_bytes_len), u8(f.qual_bytes, f.qual_bytes_len),
                    u8(f.id_bytes, f.id_bytes_len), i64(f.ends), i64(f.id_ends),
                    f.term_code, f.term_record, f.consumed, f.n_newlines, f.term_msg)
    finally:
        lib().orc_flat_free(C.byref(f))


def generate_synthetic(num_reads: int, min_len: int, max_len: int, min_phred: int, max_phred: int,
                       schema_name: str = "generic", gc_bias: float = 0.5, first: int = 0,
                       count: Optional[int] = None) -> np.ndarray:
    """generate_synthetic_fastq_buffer (utils.mojo:831-917); records [first, first+count)."""
    if count is None:
        count = num_reads - first
    L = lib()
    need = L.orc_generate_synthetic(num_reads, first, count, min_len, max_len, min_phred, max_phred,
                                    schema_name.encode(), gc_bias, None, 0)
    if need < 0:
        raise ValueError("generate_synthetic_fastq_buffer: invalid arguments")
    out = np.empty(need, dtype=np.uint8)
    if need:
        L.orc_generate_synthetic(num_reads, first, count, min_len, max_len, min_phred, max_phred,
                                 schema_name.encode(), gc_bias, out.ctypes.data, need)
    return out


def compute_num_reads_for_size(target: int, min_len: int, max_len: int) -> int:
    return lib().orc_compute_num_reads_for_size(target, min_len, max_len)


def bench_run(data: np.ndarray, config: OrcConfig, mode: str = "batches"):
    bp = C.c_int64()
    d = _as_u8(data)
    n = lib().orc_bench_run(d.ctypes.data, d.size, C.byref(config), 0 if mode == "views" else 1, C.byref(bp))
    return n, bp.value


def nw_score(ref: bytes, query: bytes) -> int:
    """examples/nw_gpu/kernels.mojo:21-89 on the CPU (orc_nw_score)."""
    return int(lib().orc_nw_score(ref, len(ref), query, len(query)))


def pipeline_run(data: np.ndarray, config: OrcConfig, ref: bytes, max_pos: int):
    """parse in batches(config.batch_size) + NW score of every record + per-position quality distribution, one thread
    (orc_pipeline_run).  Returns (records, counts[max_pos, 128], sum of the scores)."""
    d = _as_u8(data)
    counts = np.zeros((max_pos, 128), dtype=np.uint64)
    ss, bp = C.c_int64(), C.c_int64()
    n = lib().orc_pipeline_run(d.ctypes.data, d.size, C.byref(config), ref, len(ref), max_pos, counts.ctypes.data, C.byref(ss), C.byref(bp))
    return int(n), counts, int(ss.value)
