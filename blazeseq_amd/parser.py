"""Host-side mirror of the reference's FASTQ parser API over the C ABI.

Same names, argument meaning and error behaviour as the reference (paths relative to the BlazeSeq
tree):

* ``ParserConfig``       blazeseq/fastq/parser.mojo:33-74
* ``FastqParser``        blazeseq/fastq/parser.mojo:77-274 (ctors 89-145, has_more 155, next_batch 239,
                         views/records/batches 253-274, iterators 628-735)
* ``FastqBatch``         blazeseq/fastq/record_batch.mojo:19-207
* ``DeviceFastqBatch``   blazeseq/fastq/record_batch.mojo:210-244

The difference is where the work happens: a chunk of the input is parsed, validated and packed into
the FastqBatch columns on the GPU in one go, and ``batches()`` hands out zero-copy slices of those
columns; ``FastqBatch.to_device()`` therefore costs nothing.
"""
from __future__ import annotations

import ctypes as C
import io
import os
import weakref
from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import numpy as np

from . import _lib as L

DEFAULT_CAPACITY = 256 * 1024      # blazeseq/CONSTS.mojo:26
MAX_CAPACITY = 1 << 30             # blazeseq/CONSTS.mojo:28
DEFAULT_BATCH_SIZE = 4096          # blazeseq/CONSTS.mojo:31
DEFAULT_CHUNK_BYTES = 256 << 20


class ParseError(Exception):
    """The reference raises ``Error(String)``; ``code`` is the FastxErrorCode behind it
    (blazeseq/errors.mojo:33-68) and ``message`` the exact text as bytes."""

    def __init__(self, code: int, message: bytes):
        super().__init__(message.decode("latin-1"))
        self.code = code
        self.message = message


class EOFError_(ParseError):
    pass


def quality_schema(name: str) -> Tuple[int, int, int, bool]:
    """_parse_schema, blazeseq/utils.mojo:612-637 -> (LOWER, UPPER, OFFSET, known)."""
    lo, up, off = C.c_uint8(), C.c_uint8(), C.c_uint8()
    known = L.lib().bzq_schema_from_name(name.encode(), C.byref(lo), C.byref(up), C.byref(off))
    if not known:
        print("Unknown quality schema please choose one of 'sanger', 'solexa', 'illumina_1.3', 'illumina_1.5' "
              "'illumina_1.8', or 'generic'.\nParsing with generic schema.")
    return lo.value, up.value, off.value, bool(known)


@dataclass
class ParserConfig:
    """blazeseq/fastq/parser.mojo:33-74 (same fields, same defaults) plus two GPU-side switches."""
    buffer_capacity: int = DEFAULT_CAPACITY
    buffer_max_capacity: int = MAX_CAPACITY
    buffer_growth_enabled: bool = False
    check_ascii: bool = False
    check_quality: bool = False
    quality_schema: Optional[str] = None
    compat_simd_width: int = 0     # reproduce the host-SIMD-width quirk of the quality check (16 / 32 / 64);
                                   # -1 = this host's width (bzq_host_simd_width): bit-exact with the reference binary run here
    emit_offsets: bool = False     # also materialise the RecordOffsets columns
    views_only: bool = False       # views() mode: no columns, records as offsets + id spans into the chunk


def _check(ctx_handle, rc: int, what: str):
    if rc < 0:
        msg = L.lib().bzq_last_error(ctx_handle)
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    if isinstance(data, str):
        data = data.encode("latin-1")
    return np.frombuffer(bytes(data), dtype=np.uint8)


class ChunkResult:
    """bzq_chunk plus helpers that copy device columns to numpy (tests, CPU consumers)."""

    def __init__(self, ctx: "Context", raw: L.BzqChunk):
        self.ctx = ctx
        self.raw = raw
        for name, _ in L.BzqChunk._fields_:
            setattr(self, name, getattr(raw, name))
        self._serial = L.lib().bzq_set_option(ctx.h, b"n_submits", 0)   # (query) which chunk of the ctx this is

    def _i64(self, ptr, count) -> np.ndarray:
        out = np.empty(int(count), dtype=np.int64)
        if count:
            self.ctx.copy_to_host(out, ptr, int(count) * 8)
        return out

    def _u8(self, ptr, count) -> np.ndarray:
        out = np.empty(int(count), dtype=np.uint8)
        if count:
            self.ctx.copy_to_host(out, ptr, int(count))
        return out

    def _cumulative(self):
        """bzq_chunk.d_ends / d_id_ends are produced on demand (the emit kernel writes the per-batch arrays directly)."""
        if not self.d_ends and int(self.n_records):
            # the library finds the chunk's output set by its arrays and its serial (bzq_chunk.chunk_serial): the current chunk or the
            # one before it (double buffering); a chunk from two submits ago is refused there (BZQ_ERR_ARG -> RuntimeError)
            _check(self.ctx.h, L.lib().bzq_chunk_cumulative_ends(self.ctx.h, C.byref(self.raw)), "bzq_chunk_cumulative_ends")
            self.d_ends, self.d_id_ends = self.raw.d_ends, self.raw.d_id_ends

    def ends(self):
        self._cumulative()
        return self._i64(self.d_ends, self.n_records)

    def id_ends(self):
        self._cumulative()
        return self._i64(self.d_id_ends, self.n_records)
    def batch_ends(self): return self._i64(self.d_batch_ends, self.n_records)
    def batch_id_ends(self): return self._i64(self.d_batch_id_ends, self.n_records)
    def record_end(self): return self._i64(self.d_record_end, self.n_records)
    def header_start(self): return self._i64(self.d_header_start, self.n_records)
    def seq_start(self): return self._i64(self.d_seq_start, self.n_records)
    def sep_start(self): return self._i64(self.d_sep_start, self.n_records)
    def qual_start(self): return self._i64(self.d_qual_start, self.n_records)
    def id_start(self): return self._i64(self.d_id_start, self.n_records)
    def id_len(self):
        out = np.empty(int(self.n_records), dtype=np.int32)
        if out.size:
            self.ctx.copy_to_host(out, self.d_id_len, out.nbytes)
        return out
    def seq(self): return self._u8(self.d_seq, self.seq_bytes)
    def qual(self): return self._u8(self.d_qual, self.qual_bytes)
    def id(self): return self._u8(self.d_id, self.id_bytes)


class Context:
    """One bzq_ctx (= one FastqParser's device side).  Not thread safe, like the reference parser."""

    def __init__(self, config: Optional[ParserConfig] = None, schema: str = "generic",
                 batch_size: int = DEFAULT_BATCH_SIZE, device: int = 0, pass_bytes: int = 0,
                 min_record_bytes: int = 32):
        lib = L.lib()
        self.config = config if config is not None else ParserConfig()
        c = L.BzqConfig()
        lib.bzq_config_default(C.byref(c))
        c.buffer_capacity = self.config.buffer_capacity
        c.buffer_max_capacity = self.config.buffer_max_capacity
        c.buffer_growth_enabled = int(self.config.buffer_growth_enabled)
        c.check_ascii = int(self.config.check_ascii)
        c.check_quality = int(self.config.check_quality)
        # config.quality_schema overrides the ctor's schema argument (parser.mojo:134-139)
        name = self.config.quality_schema if self.config.quality_schema else schema
        c.q_lower, c.q_upper, c.q_offset, _ = quality_schema(name)
        c.batch_size = batch_size
        c.compat_simd_width = lib.bzq_host_simd_width() if self.config.compat_simd_width < 0 else self.config.compat_simd_width
        c.emit_offsets = int(self.config.emit_offsets)
        c.views_only = int(self.config.views_only)
        c.pass_bytes = pass_bytes
        c.min_record_bytes = min_record_bytes
        self.raw_config = c
        self.batch_size = batch_size
        self.quality_offset_schema = c.q_offset
        h = C.c_void_p()
        rc = lib.bzq_create(device, C.byref(c), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"bzq_create failed ({rc}): {lib.bzq_last_error(None).decode()}")
        self.h = h
        self._keep = None
        self._ingests = weakref.WeakSet()   # open Ingest objects: they use this ctx and must be closed before it

    def close(self):
        if getattr(self, "h", None):
            for ing in list(getattr(self, "_ingests", ())):
                ing.close()
            L.lib().bzq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: int):
        _check(self.h, L.lib().bzq_set_option(self.h, key.encode(), int(value)), "bzq_set_option")

    def set_stream(self, hip_stream: int):
        _check(self.h, L.lib().bzq_set_stream(self.h, C.c_void_p(hip_stream)), "bzq_set_stream")

    def submit_host(self, data, stream_pos: int = 0, is_eof: bool = True):
        arr = _as_u8(data)
        self._keep = arr
        _check(self.h, L.lib().bzq_submit_chunk_host(self.h, arr.ctypes.data if arr.size else None, arr.size,
                                                    stream_pos, int(is_eof)), "bzq_submit_chunk_host")

    def submit_device(self, d_ptr: int, n: int, stream_pos: int = 0, is_eof: bool = True):
        _check(self.h, L.lib().bzq_submit_chunk_device(self.h, C.c_void_p(d_ptr), n, stream_pos, int(is_eof)),
               "bzq_submit_chunk_device")

    def result(self) -> ChunkResult:
        raw = L.BzqChunk()
        rc = L.lib().bzq_chunk_result(self.h, C.byref(raw))
        _check(self.h, rc, "bzq_chunk_result")
        return ChunkResult(self, raw)

    def parse(self, data, stream_pos: int = 0, is_eof: bool = True) -> ChunkResult:
        self.submit_host(data, stream_pos, is_eof)
        return self.result()

    def format_error(self, records_before: int = 0) -> bytes:
        buf = C.create_string_buffer(4096)
        n = L.lib().bzq_format_error(self.h, records_before, buf, 4096)
        return buf.raw[:max(0, min(n, 4095))]

    def copy_to_host(self, out: np.ndarray, d_ptr, nbytes: int):
        _check(self.h, L.lib().bzq_copy_to_host(self.h, out.ctypes.data, C.c_void_p(d_ptr), nbytes), "bzq_copy_to_host")

    def batch_view(self, first_record: int, max_records: int) -> L.BzqDeviceBatch:
        b = L.BzqDeviceBatch()
        _check(self.h, L.lib().bzq_batch_view(self.h, first_record, max_records, C.byref(b)), "bzq_batch_view")
        return b

    def batches(self, max_records: int):
        """bzq_batches: every batch of the current chunk (a ctypes array of bzq_device_batch) in one call."""
        n = C.c_uint64()
        _check(self.h, L.lib().bzq_batches(self.h, max_records, None, 0, C.byref(n)), "bzq_batches")
        arr = (L.BzqDeviceBatch * max(1, n.value))()
        _check(self.h, L.lib().bzq_batches(self.h, max_records, arr, n.value, C.byref(n)), "bzq_batches")
        return arr, int(n.value)

    def generate_synthetic_device(self, num_reads: int, read_len: int, min_phred: int, max_phred: int,
                                  schema: str, d_out: int = 0, cap: int = 0, first: int = 0,
                                  count: Optional[int] = None, max_len: Optional[int] = None) -> int:
        """Bytes of records [first, first+count) of the reference generator's num_reads-record file; written to
        d_out when given.  ``max_len`` > ``read_len``: variable read lengths (utils.mojo:753-757)."""
        if count is None:
            count = num_reads - first
        nb = C.c_uint64()
        _check(self.h, L.lib().bzq_generate_synthetic_device_var(self.h, num_reads, first, count, read_len,
                                                                read_len if max_len is None else max_len, min_phred,
                                                                max_phred, schema.encode(),
                                                                C.c_void_p(d_out) if d_out else None, cap, C.byref(nb)),
               "bzq_generate_synthetic_device_var")
        return nb.value

    # ---- the multi-GPU protocol behind the C ABI (bzq_comm.hpp) ------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId through the C ABI: rank 0 creates it, the host hands it to the other ranks."""
        nid = L.BzqNcclId()
        rc = L.lib().bzq_comm_get_unique_id(C.byref(nid))
        if rc != 0:
            raise RuntimeError(f"bzq_comm_get_unique_id failed ({rc})")
        return bytes(nid)

    def comm_init(self, rank: int, nranks: int, nccl_id: Optional[bytes] = None):
        buf = C.create_string_buffer(nccl_id, 128) if nccl_id is not None else None
        _check(self.h, L.lib().bzq_comm_init(self.h, rank, nranks, buf), "bzq_comm_init")

    def comm_init_shm(self, rank: int, nranks: int, name: str, halo_capacity: int = 0):
        _check(self.h, L.lib().bzq_comm_init_shm(self.h, rank, nranks, name.encode(), halo_capacity), "bzq_comm_init_shm")

    def bgzf_scan(self, comp: np.ndarray, max_out: int = 1 << 62, cap: int = 0):
        """bzq_bgzf_scan over a host buffer -> (blocks, n_blocks, consumed, out_bytes)."""
        cap = cap or int(comp.size // 28 + 1)
        blocks = (L.BzqBgzfBlock * cap)()
        n, consumed, out_bytes = C.c_int64(), C.c_uint64(), C.c_uint64()
        rc = L.lib().bzq_bgzf_scan(comp.ctypes.data, comp.size, max_out, blocks, cap, C.byref(n), C.byref(consumed), C.byref(out_bytes))
        if rc < 0:
            raise RuntimeError(f"bzq_bgzf_scan: not a BGZF block at offset {int(consumed.value)}")
        return blocks, int(n.value), int(consumed.value), int(out_bytes.value)

    def bgzf_inflate(self, d_comp: int, comp_bytes: int, blocks, n_blocks: int, d_out: int, out_capacity: int):
        """bzq_bgzf_inflate: device-resident BGZF blocks -> their bytes, in device memory."""
        _check(self.h, L.lib().bzq_bgzf_inflate(self.h, C.c_void_p(d_comp), comp_bytes, blocks, n_blocks, C.c_void_p(d_out), out_capacity),
               "bzq_bgzf_inflate")

    def comm_selftest(self):
        """One checked ring exchange + all-gather over the communicator (collective)."""
        _check(self.h, L.lib().bzq_comm_selftest(self.h), "bzq_comm_selftest")

    def comm_destroy(self):
        _check(self.h, L.lib().bzq_comm_destroy(self.h), "bzq_comm_destroy")

    def shard_read_range(self, path: str, lo: int, hi: int, halo_room: int = 0, n_threads: int = 0):
        """bzq_shard_read_range: bytes [lo, hi) of a file into device memory of the ctx -> (device pointer, n, capacity) for shard_stitch."""
        p, n, cap = C.c_void_p(), C.c_uint64(), C.c_uint64()
        _check(self.h, L.lib().bzq_shard_read_range(self.h, os.fsencode(path), int(lo), int(hi), int(halo_room), int(n_threads),
                                                    C.byref(p), C.byref(n), C.byref(cap)), "bzq_shard_read_range")
        return int(p.value or 0), int(n.value), int(cap.value)

    def shard_stitch(self, d_ptr: int, n: int, capacity: int) -> "ShardResult":
        raw = L.BzqShardResult()
        _check(self.h, L.lib().bzq_shard_stitch(self.h, C.c_void_p(d_ptr), n, capacity, C.byref(raw)), "bzq_shard_stitch")
        return ShardResult(self, raw)

    def global_counts(self):
        out = (C.c_uint64 * 3)()
        _check(self.h, L.lib().bzq_global_counts(self.h, out), "bzq_global_counts")
        return [int(x) for x in out]

    def set_consumer_stream(self, hip_stream: int):
        _check(self.h, L.lib().bzq_set_consumer_stream(self.h, C.c_void_p(hip_stream)), "bzq_set_consumer_stream")

    def shard_scan(self, d_ptr: int, n: int) -> L.BzqShardSummary:
        s = L.BzqShardSummary()
        _check(self.h, L.lib().bzq_shard_scan(self.h, C.c_void_p(d_ptr), n, C.byref(s)), "bzq_shard_scan")
        return s

    def submit_shard(self, d_ptr: int, n: int, halo_bytes: int, lines_before: int, prev_last_byte: int,
                     stream_pos: int, is_last: bool):
        _check(self.h, L.lib().bzq_submit_shard(self.h, C.c_void_p(d_ptr), n, halo_bytes, lines_before,
                                               prev_last_byte, stream_pos, int(is_last)), "bzq_submit_shard")


class ShardResult:
    """bzq_shard_result: this rank's ChunkResult plus everything global the protocol derived."""

    def __init__(self, ctx: "Context", raw: L.BzqShardResult):
        self.raw = raw
        self.chunk = ChunkResult(ctx, raw.chunk)
        self.plan = raw.plan
        for name in ("stream_pos", "records_before", "global_records", "global_bases", "global_bytes", "first_error_record",
                     "stream_status", "error_rank"):
            setattr(self, name, int(getattr(raw, name)))


@dataclass
class FastqRecord:
    """Owned record (blazeseq/fastq/record.mojo:230-428), plumbing only."""
    id: bytes
    sequence: bytes
    quality: bytes
    phred_offset: int = 33

    def __len__(self):
        return len(self.sequence)


class FastqView:
    """``FastqView`` (blazeseq/fastq/record.mojo:431-550): three spans into the parser's buffer, valid until the parser
    moves to its next chunk (the reference: until the next parser call).  The spans are ``memoryview`` slices of the
    host copy of the chunk; WHERE they are comes from the device (views mode, csrc/bzq_views.hpp)."""
    __slots__ = ("id", "sequence", "quality", "phred_offset")

    def __init__(self, id, sequence, quality, phred_offset=33):
        self.id, self.sequence, self.quality, self.phred_offset = id, sequence, quality, phred_offset

    def __len__(self):   # record.mojo:474-476: the sequence length
        return len(self.sequence)

    def byte_len(self) -> int:   # record.mojo:478-487
        return 1 + len(self.id) + len(self.sequence) + len(self.quality) + 5

    def to_record(self) -> FastqRecord:
        return FastqRecord(bytes(self.id), bytes(self.sequence), bytes(self.quality), self.phred_offset)


class DeviceFastqBatch:
    """blazeseq/fastq/record_batch.mojo:210-220: five device buffers + four scalars.  The buffers are
    device pointers (ints) into the parser's chunk columns."""

    def __init__(self, ctx: Context, raw: L.BzqDeviceBatch, owned: bool = False):
        self._ctx = ctx
        self.raw = raw
        self._owned = owned   # uploaded with bzq_upload_batch: this object frees the device memory
        self.num_records = raw.num_records
        self.seq_len = raw.seq_len
        self.quality_offset = raw.quality_offset
        self.total_id_bytes = raw.total_id_bytes
        self.qual_buffer = raw.qual_buffer
        self.sequence_buffer = raw.sequence_buffer
        self.ends = raw.ends
        self.id_buffer = raw.id_buffer
        self.id_ends = raw.id_ends

    def copy_to_host(self) -> "FastqBatch":
        """record_batch.mojo:222-244"""
        b = FastqBatch(self._ctx, self.raw)
        if self._owned:
            b._detach()   # the host copy must not depend on this object's lifetime
        return b

    def release(self):
        if self._owned and getattr(self._ctx, "h", None):
            L.lib().bzq_release_batch(self._ctx.h, C.byref(self.raw))
        self._owned = False

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def to_records(self) -> List[FastqRecord]:
        return self.copy_to_host().to_records()

    # ---- device-side consumers (bzq_consumers.hpp): outputs are device pointers the caller owns ----------------
    def nw_scores(self, reference: bytes, d_scores: int):
        """examples/nw_gpu: score of every record against ``reference`` into device int32[num_records]."""
        ref = bytes(reference)
        _check(self._ctx.h, L.lib().bzq_batch_nw_scores(self._ctx.h, C.byref(self.raw), ref, len(ref), C.c_void_p(d_scores)),
               "bzq_batch_nw_scores")

    def quality_sums(self, d_sums: int):
        """Per-record sum of Phred scores into device int64[num_records] (asynchronous on the ctx stream)."""
        _check(self._ctx.h, L.lib().bzq_batch_quality_sums(self._ctx.h, C.byref(self.raw), C.c_void_p(d_sums)),
               "bzq_batch_quality_sums")

    def quality_by_position(self, max_positions: int) -> np.ndarray:
        """counts[p, v]: records whose quality byte at read position p is v (uint64 [max_positions, 128])."""
        out = np.zeros((int(max_positions), 128), dtype=np.uint64)
        _check(self._ctx.h, L.lib().bzq_batch_quality_by_position(self._ctx.h, C.byref(self.raw), int(max_positions),
                                                                  out.ctypes.data_as(C.POINTER(C.c_uint64))), "bzq_batch_quality_by_position")
        return out

    def histogram(self, column: str = "sequence") -> np.ndarray:
        """256-bin byte histogram of the sequence or quality column of this batch."""
        ptr = self.sequence_buffer if column == "sequence" else self.qual_buffer
        out = (C.c_uint64 * 256)()
        _check(self._ctx.h, L.lib().bzq_column_histogram(self._ctx.h, C.c_void_p(ptr), int(self.seq_len), out),
               "bzq_column_histogram")
        return np.frombuffer(out, dtype=np.uint64).copy()


class FastqBatch:
    """blazeseq/fastq/record_batch.mojo:19-207.  Created by the parser from a device batch; the host
    copies of the five columns (``_id_bytes, _quality_bytes, _sequence_bytes, _id_ends, _ends``) are
    fetched on first use."""

    def __init__(self, ctx: Context, raw: L.BzqDeviceBatch):
        self._ctx = ctx
        self._raw = raw
        self._host = None
        self._device_valid = True
        self._quality_offset = raw.quality_offset

    def _detach(self):
        """The parser is about to recycle the chunk columns: take the host copy now, so that the
        batch stays an owned object like the reference's FastqBatch."""
        if self._device_valid and self._raw.num_records:
            self._fetch()
        self._device_valid = False

    def _fetch(self):
        if self._host is None:
            r = self._raw
            n = int(r.num_records)
            q = np.empty(int(r.seq_len), dtype=np.uint8)
            s = np.empty(max(0, int(r.sequence_bytes)), dtype=np.uint8)   # == seq_len but for an odd unterminated last record
            i = np.empty(int(r.total_id_bytes), dtype=np.uint8)
            e = np.empty(n, dtype=np.int64)
            ie = np.empty(n, dtype=np.int64)
            hb = L.BzqHostBatch()
            hb.quality_bytes = q.ctypes.data; hb.sequence_bytes = s.ctypes.data; hb.id_bytes = i.ctypes.data
            hb.ends = e.ctypes.data; hb.id_ends = ie.ctypes.data
            _check(self._ctx.h, L.lib().bzq_batch_to_host(self._ctx.h, C.byref(r), C.byref(hb)), "bzq_batch_to_host")
            self._host = (i, q, s, ie, e)
        return self._host

    @property
    def _id_bytes(self): return self._fetch()[0]
    @property
    def _quality_bytes(self): return self._fetch()[1]
    @property
    def _sequence_bytes(self): return self._fetch()[2]
    @property
    def _id_ends(self): return self._fetch()[3]
    @property
    def _ends(self): return self._fetch()[4]

    def num_records(self) -> int: return int(self._raw.num_records)
    def seq_len(self) -> int: return int(self._raw.seq_len)
    def quality_offset(self) -> int: return self._quality_offset
    def __len__(self): return self.num_records()
    def __repr__(self): return f"FastqBatch(records={self.num_records()}, quality_offset={self._quality_offset})"

    def to_device(self, ctx=None) -> DeviceFastqBatch:
        """record_batch.mojo:89-90.  While the batch's chunk is live the columns already sit on the device: zero copy,
        valid until the parser refills.  A batch kept past a refill owns a host copy and is uploaded."""
        if not self._device_valid:
            # the parser has moved past this batch's chunk: upload the owned host copy (what the reference always does)
            i, q, s, ie, e = self._fetch()
            hb = L.BzqHostBatch()
            hb.num_records = self.num_records()
            hb.quality_bytes = q.ctypes.data; hb.sequence_bytes = s.ctypes.data; hb.id_bytes = i.ctypes.data
            hb.ends = e.ctypes.data; hb.id_ends = ie.ctypes.data
            hb.quality_offset = self._quality_offset
            out = L.BzqDeviceBatch()
            _check(self._ctx.h, L.lib().bzq_upload_batch(self._ctx.h, C.byref(hb), C.byref(out)), "bzq_upload_batch")
            return DeviceFastqBatch(self._ctx, out, owned=True)
        return DeviceFastqBatch(self._ctx, self._raw)

    def get_record(self, index: int) -> FastqRecord:
        """record_batch.mojo:116-150"""
        n = self.num_records()
        if index < 0 or index >= n:
            raise IndexError("FastqBatch.get_record index out of range")
        i, q, s, ie, e = self._fetch()
        i0 = 0 if index == 0 else int(ie[index - 1])
        s0 = 0 if index == 0 else int(e[index - 1])
        return FastqRecord(i[i0:int(ie[index])].tobytes(), s[s0:int(e[index])].tobytes(),
                           q[s0:int(e[index])].tobytes(), self._quality_offset)

    get_ref = get_record

    def to_records(self) -> List[FastqRecord]:
        return [self.get_record(k) for k in range(self.num_records())]


class GzipDecoder:
    """bzq_gzip_*: any gzip stream inflated on the device in parallel (the reference's RapidgzipReader / GZFile,
    blazeseq/io/readers.mojo:283-443).  A stream decoder: feed() the file piece by piece, in order."""

    def __init__(self, ctx: Context, chunk_bytes: int = 0):
        self._ctx = ctx
        self._h = C.c_void_p()
        _check(ctx.h, L.lib().bzq_gzip_open(ctx.h, C.byref(self._h)), "bzq_gzip_open")
        if chunk_bytes:
            self._gcheck(L.lib().bzq_gzip_set_option(self._h, b"chunk_bytes", chunk_bytes), "bzq_gzip_set_option")

    def _gcheck(self, rc: int, what: str):
        if rc < 0:
            raise RuntimeError(f"{what} failed ({rc}): {(L.lib().bzq_gzip_last_error(self._h) or b'').decode('latin-1')}")

    def set_option(self, key: str, value: int) -> int:
        """bzq_gzip_set_option: "chunk_bytes", "host_continuation" (default 1: a stretch without block starts the finder can find --
        fixed-Huffman / stored blocks only -- is continued by zlib on the host), "far_kib"; query "host_calls"."""
        rc = L.lib().bzq_gzip_set_option(self._h, key.encode(), int(value))
        self._gcheck(rc, "bzq_gzip_set_option")
        return rc

    def feed(self, comp, is_last: bool, d_out: int, out_capacity: int):
        """-> (bytes written at d_out, more): `more` = out_capacity cut the output short, or the next stretch goes to the host: call
        again (an empty piece is fine)."""
        a = _as_u8(comp)
        nb, more = C.c_uint64(), C.c_int32()
        self._gcheck(L.lib().bzq_gzip_decode(self._h, a.ctypes.data if a.size else None, a.size, int(is_last), C.c_void_p(d_out), out_capacity,
                                             C.byref(nb), C.byref(more)), "bzq_gzip_decode")
        return int(nb.value), bool(more.value)

    def stage(self, comp):
        """Read-ahead (bzq_gzip_stage): the piece AFTER the one about to be fed starts its way to the device; feed() of the
        same array later finds it there.  `comp`: pinned memory (Context.pinned_array), untouched until that feed() returns."""
        a = _as_u8(comp)
        if a.size:
            L.lib().bzq_gzip_stage(self._h, a.ctypes.data, a.size)   # (a hint: a piece that was not staged is copied by its feed())

    @property
    def finished(self) -> bool:
        return bool(L.lib().bzq_gzip_finished(self._h))

    def stats(self) -> L.BzqGzipStats:
        st = L.BzqGzipStats()
        self._gcheck(L.lib().bzq_gzip_get_stats(self._h, C.byref(st)), "bzq_gzip_get_stats")
        return st

    def close(self):
        if self._h:
            L.lib().bzq_gzip_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ingest:
    """bzq_ingest: file -> pinned double buffers -> device -> chunk parser (native reader threads, the H2D copy of
    chunk k+1 overlaps the consumption of chunk k).  ``next(records_taken)`` parses the next chunk; the records of
    the previous chunk that were not taken are carried in front of it."""

    def __init__(self, ctx: Context, path: str, chunk_bytes: int = 0, n_threads: int = 0):
        self._ctx = ctx
        h = C.c_void_p()
        _check(ctx.h, L.lib().bzq_ingest_open(ctx.h, os.fsencode(path), int(chunk_bytes), int(n_threads), C.byref(h)),
               "bzq_ingest_open")
        self.h = h
        self.stream_pos = 0
        ctx._ingests.add(self)

    def next(self, records_taken: int = 0) -> ChunkResult:
        raw = L.BzqChunk()
        sp = C.c_uint64(0)
        rc = L.lib().bzq_ingest_next(self.h, int(records_taken), C.byref(raw), C.byref(sp))
        _check(self._ctx.h, rc, "bzq_ingest_next")
        self.stream_pos = int(sp.value)
        return ChunkResult(self._ctx, raw)

    def stats(self) -> L.BzqIngestStats:
        st = L.BzqIngestStats()
        L.lib().bzq_ingest_get_stats(self.h, C.byref(st))
        return st

    def close(self):
        if getattr(self, "h", None):
            L.lib().bzq_ingest_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Source:
    """Reader.read_to_buffer semantics (blazeseq/io/readers.mojo:51-79): returns up to n bytes, b"" at EOF."""

    def __init__(self, src):
        self._mv = None
        self._f = None
        self._pos = 0
        if isinstance(src, str) and src and "\n" not in src and not src.startswith("@") and not os.path.exists(src):
            # a str with no newline that does not start a FASTQ record is a (mistyped) path, not content
            raise FileNotFoundError(f"FastqParser: no such file: {src!r} (pass bytes for in-memory FASTQ content)")
        if isinstance(src, (bytes, bytearray, memoryview, np.ndarray, str)) and not (isinstance(src, str) and os.path.exists(src)):
            self._mv = _as_u8(src)
        elif isinstance(src, (str, os.PathLike)):
            self._f = open(src, "rb", buffering=0)
        elif hasattr(src, "read"):
            self._f = src
        else:
            raise TypeError("FastqParser source must be bytes, a numpy array, a path or a binary file object")

    def read(self, n: int) -> np.ndarray:
        if self._mv is not None:
            out = self._mv[self._pos:self._pos + n]
            self._pos += out.size
            return out
        b = self._f.read(n)
        return np.frombuffer(b if b else b"", dtype=np.uint8)


class FastqParser:
    """blazeseq/fastq/parser.mojo:77-274 over GPU chunks.

    ``FastqParser(source, schema="generic", batch_size=4096, config=ParserConfig())`` -- the three
    reference constructors collapse into keyword arguments; ``config.quality_schema`` overrides
    ``schema`` (parser.mojo:134-139).  ``source``: bytes / numpy uint8 / path / binary file object.
    """

    def __init__(self, source, schema: str = "generic", batch_size: int = DEFAULT_BATCH_SIZE,
                 config: Optional[ParserConfig] = None, device: int = 0, chunk_bytes: int = DEFAULT_CHUNK_BYTES,
                 pass_bytes: int = 0, native_ingest: bool = True, reader_threads: int = 0, gpu_inflate: bool = True):
        self.config = config if config is not None else ParserConfig()
        self._batch_size = batch_size
        self._ctx = Context(self.config, schema, batch_size, device, pass_bytes=pass_bytes)
        if not gpu_inflate:   # BGZF files: inflate on the reader threads instead of on the device
            self._ctx.set_option("ingest_gpu_inflate", 0)
        # a plain file goes through the native ingest pipeline (reader threads + pinned double buffers);
        # bytes / arrays / file objects through the Reader.read_to_buffer-style loop below
        self._ingest: Optional[Ingest] = None
        if native_ingest and not self.config.views_only and isinstance(source, (str, os.PathLike)) and os.path.isfile(source):
            self._ingest = Ingest(self._ctx, os.fspath(source), max(int(chunk_bytes), 1 << 16), reader_threads)
            self._src = None
        else:
            self._src = _Source(source)
        self._taken = 0               # records of the previous ingest chunk that were handed out
        self._chunk_bytes = max(int(chunk_bytes), 1 << 16)
        self._carry = np.zeros(0, dtype=np.uint8)
        self._stream_pos = 0          # stream offset of the current chunk's first byte
        self._records_before = 0      # records delivered by earlier chunks
        self._src_eof = False
        self._chunk: Optional[ChunkResult] = None
        self._chunk_data: Optional[np.ndarray] = None
        self._next = 0                # next record of the current chunk to hand out
        self._terminal: Optional[Tuple[int, bytes]] = None  # (code, message) once the stream has ended
        self._eof_seen = False
        self._live = weakref.WeakSet()   # batches handed out from the current chunk

    # ------------------------------------------------------------------ chunk pipeline
    def _load_chunk(self, min_records: int):
        """Read + parse the next chunk (carry first).  Grows the chunk until it holds at least
        ``min_records`` complete records or the stream ends."""
        if self._ingest is not None:
            while True:
                res = self._ingest.next(self._taken)
                self._taken = 0
                if res.status == L.OK and int(res.n_records) < min_records:
                    continue   # carry everything and append the next piece of the file
                break
            self._stream_pos = self._ingest.stream_pos
            self._chunk, self._chunk_data, self._next = res, None, 0
            if res.status != L.OK:
                self._terminal = (int(res.status), self._ctx.format_error(self._records_before))
            return
        want = max(self._chunk_bytes, self._carry.size + (1 << 16))
        data = self._carry
        while True:
            parts = [data]
            have = data.size
            while have < want and not self._src_eof:
                blk = self._src.read(want - have)
                if blk.size == 0:
                    self._src_eof = True
                    break
                parts.append(blk)
                have += blk.size
            if len(parts) > 1:
                data = np.concatenate(parts)
            # records delivered so far: the library keeps the stream's record ends so that the tail of a stream parsed in
            # several chunks is judged with the reference's window where it really is (io/buffered.mojo:239-290)
            self._ctx.set_option("records_before", self._records_before)
            self._ctx.submit_host(data, self._stream_pos, self._src_eof)
            res = self._ctx.result()
            if res.status == L.OK and int(res.n_records) < min_records:
                want = max(want * 2, data.size * 2)   # not the end of the stream: read further
                continue
            break
        self._carry = np.zeros(0, dtype=np.uint8)
        self._chunk, self._chunk_data, self._next = res, data, 0
        if res.status != L.OK:
            self._terminal = (int(res.status), self._ctx.format_error(self._records_before))

    def _retire_chunk(self):
        """Drop the current chunk; bytes from the first record not handed out become the carry
        (the chunk-level analogue of the SearchPhase resume, utils.mojo:485-487)."""
        for b in list(self._live):
            b._detach()
        self._live = weakref.WeakSet()
        res, data = self._chunk, self._chunk_data
        if self._ingest is not None:
            self._taken = self._next
            self._records_before += self._next
            self._chunk = None
            return
        if self._next == 0:
            cut = 0
        elif self._next == int(res.n_records):
            cut = int(res.bytes_consumed)
        else:
            one = np.empty(1, dtype=np.int64)
            self._ctx.copy_to_host(one, res.d_record_end + 8 * (self._next - 1), 8)
            cut = int(one[0]) + 1
        self._carry = data[cut:].copy()
        self._stream_pos += cut
        self._records_before += self._next
        self._chunk = None
        self._chunk_data = None

    # ------------------------------------------------------------------ reference API
    def has_more(self) -> bool:
        """parser.mojo:155-157: True until the end of the stream has been observed."""
        return not self._eof_seen

    def next_batch(self, max_records: int = DEFAULT_BATCH_SIZE, partial_ok: bool = False) -> FastqBatch:
        """parser.mojo:239-251.  Up to ``max_records`` records; fewer (possibly zero) only at the end
        of the stream; raises ParseError when the batch would reach the stream's failing record
        (the records already collected for that batch are dropped, exactly like the reference's
        raise out of ``batch.add(self.next_view())``).  ``partial_ok`` (record-wise iteration in bulk): the records
        before a failing record are returned first, the error is raised by the following call."""
        limit = max_records if max_records else self._batch_size
        while True:
            if self._chunk is None:
                if self._terminal is not None:
                    break
                self._load_chunk(limit)
            avail = int(self._chunk.n_records) - self._next
            if avail >= limit or self._terminal is not None:
                break
            self._retire_chunk()   # not enough records left and more input follows: re-chunk
        avail = (int(self._chunk.n_records) - self._next) if self._chunk is not None else 0
        if avail < limit:
            # the batch runs into the terminal event of the stream
            code, msg = self._terminal
            if code != L.EOF and partial_ok and avail > 0:
                limit = avail   # hand out what precedes the failing record; the next call raises
            else:
                self._eof_seen = True
                if code != L.EOF:
                    self._next += avail
                    raise ParseError(code, msg)
        take = min(limit, avail)
        if take == 0:
            return FastqBatch(self._ctx, L.BzqDeviceBatch())
        raw = self._ctx.batch_view(self._next, take)
        self._next += take
        b = FastqBatch(self._ctx, raw)
        self._live.add(b)
        return b

    def batches(self, max_records: Optional[int] = None) -> Iterator[FastqBatch]:
        """parser.mojo:267-274 + _FastqParserBatchIter 700-735: stops on an empty batch or on any
        error (printing it when it carries a record number)."""
        limit = max_records if max_records else self._batch_size
        while self.has_more():
            try:
                b = self.next_batch(limit)
            except ParseError as e:
                if b"Record number:" in e.message:
                    print(e.message.decode("latin-1"))
                return
            if len(b) == 0:
                return
            yield b

    def next_record(self) -> FastqRecord:
        """parser.mojo:188-211 (config-1 plumbing: one record at a time through a 1-record batch)."""
        b = self.next_batch(1)
        if len(b) == 0:
            raise EOFError_(L.EOF, b"EOF")
        r = b.get_record(0)
        r.phred_offset = self._ctx.quality_offset_schema
        return r

    # ------------------------------------------------------------------ views (parser.mojo:159-170, 253-258, 628-661)
    def _view_chunk(self):
        """Host copies of the current chunk's RecordOffsets / id spans (views mode: the device wrote nothing else)."""
        res = self._chunk
        if getattr(res, "_view_cols", None) is None:
            res._view_cols = (res.seq_start(), res.sep_start(), res.qual_start(), res.record_end(), res.id_start(), res.id_len(),
                              memoryview(self._chunk_data))
        return res._view_cols

    def next_view(self) -> "FastqView":
        """parser.mojo:159-170.  With ``config.views_only`` the spans are zero-copy slices of the chunk at offsets the
        views-mode kernels produced; otherwise a view of a 1-record batch (plumbing)."""
        if not self.config.views_only:
            r = self.next_record()
            return FastqView(r.id, r.sequence, r.quality, r.phred_offset)
        while True:
            if self._chunk is None:
                if self._terminal is not None:
                    break
                self._load_chunk(1)
            if self._next < int(self._chunk.n_records):
                ss, sp, qs, re, ids, idl, mv = self._view_chunk()
                r = self._next
                self._next += 1
                return FastqView(mv[ids[r]:ids[r] + idl[r]], mv[ss[r]:sp[r] - 1], mv[qs[r]:re[r]], self._ctx.quality_offset_schema)
            if self._terminal is not None:
                break
            self._retire_chunk()
        self._eof_seen = True
        code, msg = self._terminal
        raise (EOFError_ if code == L.EOF else ParseError)(code, msg if code != L.EOF else b"EOF")

    def views(self) -> Iterator["FastqView"]:
        """parser.mojo:253-258 + _FastqParserViewIter 628-661: EOF ends the iteration, any other error is printed first."""
        if not self.config.views_only:
            for r in self.records():
                yield FastqView(r.id, r.sequence, r.quality, r.phred_offset)
            return
        while True:
            try:
                yield self.next_view()
            except EOFError_:
                return
            except ParseError as e:
                print(e.message.decode("latin-1"))
                return

    def records(self) -> Iterator[FastqRecord]:
        """parser.mojo:260-265 + _FastqParserRecordIter 664-697: errors are printed and end the iteration."""
        while self.has_more():
            try:
                b = self.next_batch(self._batch_size, partial_ok=True)   # the records before a failing one still come out
            except ParseError as e:
                print(e.message.decode("latin-1"))
                return
            if len(b) == 0:
                return
            for r in b.to_records():
                r.phred_offset = self._ctx.quality_offset_schema
                yield r

