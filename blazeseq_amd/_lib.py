"""ctypes binding of libblazeseq_hip.so (the C ABI declared in include/blazeseq_hip.h).

The HIP library IS the product path: if it is missing or cannot be loaded this module raises --
there is no CPU fallback and nothing here ever imports the test oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BLAZESEQ_HIP_LIB") or os.path.join(_HERE, "libblazeseq_hip.so")  # override: A/B of two builds

OK, ID_NO_AT, SEP_NO_PLUS, SEQ_QUAL_LEN_MISMATCH, ASCII_INVALID, QUALITY_OUT_OF_RANGE, EOF, \
    UNEXPECTED_EOF, BUFFER_EXCEEDED, BUFFER_AT_MAX, OTHER = range(11)
ERR_HIP, ERR_ARG, ERR_NOMEM, ERR_NO_DEVICE, ERR_IO = -1, -2, -3, -4, -5


class BzqConfig(C.Structure):
    _fields_ = [
        ("buffer_capacity", C.c_int64),
        ("buffer_max_capacity", C.c_int64),
        ("buffer_growth_enabled", C.c_int32),
        ("check_ascii", C.c_int32),
        ("check_quality", C.c_int32),
        ("q_lower", C.c_uint8), ("q_upper", C.c_uint8), ("q_offset", C.c_uint8), ("_pad0", C.c_uint8),
        ("batch_size", C.c_int32),
        ("compat_simd_width", C.c_int32),
        ("emit_offsets", C.c_int32),
        ("views_only", C.c_int32),
        ("max_chunk_bytes", C.c_int64),
        ("pass_bytes", C.c_int64),
        ("min_record_bytes", C.c_int32),
        ("_pad2", C.c_int32),
    ]


class BzqChunk(C.Structure):
    _fields_ = [
        ("n_bytes", C.c_uint64), ("n_records", C.c_uint64), ("bytes_consumed", C.c_uint64),
        ("total_newlines", C.c_uint64),
        ("status", C.c_int32), ("tail_phase", C.c_int32),
        ("error_record", C.c_int64),
        ("seq_bytes", C.c_uint64), ("qual_bytes", C.c_uint64), ("id_bytes", C.c_uint64),
        ("d_seq", C.c_void_p), ("d_qual", C.c_void_p), ("d_id", C.c_void_p),
        ("d_ends", C.c_void_p), ("d_id_ends", C.c_void_p),
        ("d_batch_ends", C.c_void_p), ("d_batch_id_ends", C.c_void_p),
        ("d_record_end", C.c_void_p),
        ("d_header_start", C.c_void_p), ("d_seq_start", C.c_void_p), ("d_sep_start", C.c_void_p),
        ("d_qual_start", C.c_void_p),
        ("ms_total", C.c_float), ("ms_aggregate", C.c_float), ("ms_scan", C.c_float),
        ("ms_emit", C.c_float), ("ms_rebase", C.c_float),
        ("n_passes", C.c_uint32), ("chunk_serial", C.c_uint32),
        ("d_id_start", C.c_void_p), ("d_id_len", C.c_void_p),
    ]


class BzqDeviceBatch(C.Structure):
    _fields_ = [
        ("num_records", C.c_int64), ("seq_len", C.c_int64), ("total_id_bytes", C.c_int64),
        ("quality_offset", C.c_uint8), ("_pad", C.c_uint8 * 7),
        ("qual_buffer", C.c_void_p), ("sequence_buffer", C.c_void_p), ("ends", C.c_void_p),
        ("id_buffer", C.c_void_p), ("id_ends", C.c_void_p),
        ("first_record", C.c_uint64),
        ("sequence_bytes", C.c_int64),
    ]


class BzqDeviceViews(C.Structure):
    _fields_ = [
        ("num_records", C.c_int64), ("chunk", C.c_void_p),
        ("header_start", C.c_void_p), ("seq_start", C.c_void_p), ("sep_start", C.c_void_p), ("qual_start", C.c_void_p),
        ("record_end", C.c_void_p), ("id_start", C.c_void_p), ("id_len", C.c_void_p),
        ("first_record", C.c_uint64),
    ]


class BzqHostBatch(C.Structure):
    _fields_ = [
        ("num_records", C.c_int64),
        ("quality_bytes", C.c_void_p), ("sequence_bytes", C.c_void_p), ("id_bytes", C.c_void_p),
        ("ends", C.c_void_p), ("id_ends", C.c_void_p),
        ("quality_offset", C.c_uint8), ("_pad", C.c_uint8 * 7),
    ]


class BzqShardSummary(C.Structure):
    _fields_ = [
        ("n_bytes", C.c_uint64), ("n_newlines", C.c_uint64),
        ("first_nl", C.c_int64 * 4),
        ("first_byte", C.c_uint8), ("last_byte", C.c_uint8), ("_pad", C.c_uint8 * 6),
    ]


# every symbol include/blazeseq_hip.h declares (tests/test_abi_symbols.py checks the header against this)
class BzqIngestStats(C.Structure):
    _fields_ = [
        ("file_bytes", C.c_uint64), ("bytes_read", C.c_uint64), ("chunks", C.c_uint64), ("records", C.c_uint64),
        ("read_s", C.c_double), ("wait_s", C.c_double), ("total_s", C.c_double),
        ("direct_io", C.c_int32), ("numa_node", C.c_int32),
    ]


class BzqShardPlan(C.Structure):
    _fields_ = [
        ("lines_before", C.c_uint64), ("head_bytes", C.c_uint64), ("halo_bytes", C.c_uint64), ("halo_offset", C.c_uint64),
        ("head_dst", C.c_int32), ("halo_first_src", C.c_int32), ("halo_n_src", C.c_int32),
        ("prev_last_byte", C.c_uint8), ("is_last", C.c_uint8), ("_pad", C.c_uint8 * 2),
    ]


class BzqShardResult(C.Structure):
    _fields_ = [
        ("chunk", BzqChunk), ("plan", BzqShardPlan),
        ("stream_pos", C.c_uint64), ("records_before", C.c_uint64),
        ("global_records", C.c_uint64), ("global_bases", C.c_uint64), ("global_bytes", C.c_uint64),
        ("first_error_record", C.c_int64), ("stream_status", C.c_int32), ("error_rank", C.c_int32),
    ]


class BzqNcclId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class BzqFastaConfig(C.Structure):
    _fields_ = [("check_ascii", C.c_int32), ("_pad", C.c_int32), ("line_capacity", C.c_int64)]


class BzqFastaChunk(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("_pad", C.c_int32),
        ("n_records", C.c_int64), ("bytes_consumed", C.c_uint64), ("lines_consumed", C.c_int64),
        ("seq_bytes", C.c_int64), ("id_bytes", C.c_int64),
        ("d_seq_bytes", C.c_void_p), ("d_id_bytes", C.c_void_p), ("d_seq_ends", C.c_void_p), ("d_id_ends", C.c_void_p),
        ("d_hdr_pos", C.c_void_p),
        ("err_record_number", C.c_int64), ("err_line_number", C.c_int64), ("err_file_position", C.c_int64),
        ("kernel_ms", C.c_double),
    ]


class BzqBgzfBlock(C.Structure):
    _fields_ = [("comp_offset", C.c_uint64), ("comp_size", C.c_uint32), ("out_size", C.c_uint32), ("crc32", C.c_uint32), ("_pad", C.c_uint32),
                ("out_offset", C.c_uint64)]


class BzqFastaShardSummary(C.Structure):
    _fields_ = [("n_bytes", C.c_uint64), ("first_header", C.c_int64), ("lead_kind", C.c_int32), ("last_byte", C.c_int32),
                ("tail_open", C.c_int64)]


class BzqFastaShardPlan(C.Structure):
    _fields_ = [("stream_pos", C.c_uint64), ("head_bytes", C.c_uint64), ("halo_bytes", C.c_uint64), ("halo_offset", C.c_uint64),
                ("head_dst", C.c_int32), ("halo_first_src", C.c_int32), ("halo_n_src", C.c_int32), ("is_last", C.c_int32)]


class BzqFastaShardResult(C.Structure):
    _fields_ = [("chunk", BzqFastaChunk), ("plan", BzqFastaShardPlan), ("records_before", C.c_uint64),
                ("global_records", C.c_uint64), ("first_error_record", C.c_int64), ("stream_status", C.c_int32),
                ("error_rank", C.c_int32)]


class BzqGzipStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("pieces", "bytes_in", "bytes_consumed", "bytes_out", "chunks", "chunks_with_start", "chain_jobs",
                                          "fallback_jobs", "members", "pool_retries")]


FASTA_NO_HEADER, FASTA_EMPTY_SEQUENCE, FASTA_NEED_MORE = 1, 11, 12

SYMBOLS = {
    "bzq_abi_version": (C.c_int32, []),
    "bzq_config_default": (None, [C.POINTER(BzqConfig)]),
    "bzq_schema_from_name": (C.c_int32, [C.c_char_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
    "bzq_message_for_code": (C.c_char_p, [C.c_int32]),
    "bzq_host_simd_width": (C.c_int32, []),
    "bzq_create": (C.c_int32, [C.c_int32, C.POINTER(BzqConfig), C.POINTER(C.c_void_p)]),
    "bzq_destroy": (None, [C.c_void_p]),
    "bzq_last_error": (C.c_char_p, [C.c_void_p]),
    "bzq_set_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "bzq_set_consumer_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "bzq_get_config": (C.c_int32, [C.c_void_p, C.POINTER(BzqConfig)]),
    "bzq_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    "bzq_pinned_alloc": (C.c_int32, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "bzq_pinned_free": (C.c_int32, [C.c_void_p]),
    "bzq_device_alloc": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "bzq_device_free": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "bzq_copy_to_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bzq_submit_chunk_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32]),
    "bzq_submit_chunk_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32]),
    "bzq_chunk_result": (C.c_int32, [C.c_void_p, C.POINTER(BzqChunk)]),
    "bzq_batch_view": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(BzqDeviceBatch)]),
    "bzq_batches": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(BzqDeviceBatch), C.c_uint64, C.POINTER(C.c_uint64)]),
    "bzq_chunk_cumulative_ends": (C.c_int32, [C.c_void_p, C.POINTER(BzqChunk)]),
    "bzq_views": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(BzqDeviceViews)]),
    "bzq_batch_to_host": (C.c_int32, [C.c_void_p, C.POINTER(BzqDeviceBatch), C.POINTER(BzqHostBatch)]),
    "bzq_copy_to_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bzq_format_error": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]),
    "bzq_shard_scan": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(BzqShardSummary)]),
    "bzq_submit_shard": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint8,
                                      C.c_uint64, C.c_int32]),
    "bzq_shard_head_bytes": (C.c_int32, [C.POINTER(BzqShardSummary), C.c_uint64, C.c_uint8, C.POINTER(C.c_uint64)]),
    "bzq_comm_get_unique_id": (C.c_int32, [C.POINTER(BzqNcclId)]),
    "bzq_comm_init": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "bzq_comm_init_shm": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_uint64]),
    "bzq_comm_selftest": (C.c_int32, [C.c_void_p]),
    "bzq_comm_destroy": (C.c_int32, [C.c_void_p]),
    "bzq_plan_shards": (C.c_int32, [C.POINTER(BzqShardSummary), C.c_int32, C.POINTER(BzqShardPlan)]),
    "bzq_shard_stitch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(BzqShardResult)]),
    "bzq_shard_read_range": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bzq_global_counts": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "bzq_generate_synthetic_device": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                                  C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_uint64,
                                                  C.POINTER(C.c_uint64)]),
    "bzq_generate_synthetic_device_var": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                                      C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_uint64,
                                                      C.POINTER(C.c_uint64)]),
    "bzq_ingest_open": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]),
    "bzq_ingest_next": (C.c_int32, [C.c_void_p, C.c_uint64, C.POINTER(BzqChunk), C.POINTER(C.c_uint64)]),
    "bzq_ingest_get_stats": (C.c_int32, [C.c_void_p, C.POINTER(BzqIngestStats)]),
    "bzq_ingest_close": (None, [C.c_void_p]),
    "bzq_upload_batch": (C.c_int32, [C.c_void_p, C.POINTER(BzqHostBatch), C.POINTER(BzqDeviceBatch)]),
    "bzq_release_batch": (C.c_int32, [C.c_void_p, C.POINTER(BzqDeviceBatch)]),
    "bzq_batch_nw_scores": (C.c_int32, [C.c_void_p, C.POINTER(BzqDeviceBatch), C.c_char_p, C.c_int32, C.c_void_p]),
    "bzq_batch_quality_sums": (C.c_int32, [C.c_void_p, C.POINTER(BzqDeviceBatch), C.c_void_p]),
    "bzq_batch_quality_by_position": (C.c_int32, [C.c_void_p, C.POINTER(BzqDeviceBatch), C.c_int32, C.POINTER(C.c_uint64)]),
    "bzq_batch_nw_scores_dev": (C.c_int32, [C.c_void_p, C.POINTER(BzqDeviceBatch), C.c_void_p, C.c_int32, C.c_void_p]),
    "bzq_batch_quality_by_position_acc": (C.c_int32, [C.c_void_p, C.POINTER(BzqDeviceBatch), C.c_int32, C.c_void_p]),
    "bzq_consumer_synchronize": (C.c_int32, [C.c_void_p]),
    "bzq_column_histogram": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bzq_column_gc_counts": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "bzq_fasta_create": (C.c_int32, [C.c_int32, C.POINTER(BzqFastaConfig), C.POINTER(C.c_void_p)]),
    "bzq_fasta_destroy": (None, [C.c_void_p]),
    "bzq_fasta_last_error": (C.c_char_p, [C.c_void_p]),
    "bzq_fasta_parse": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64,
                                    C.POINTER(BzqFastaChunk)]),
    "bzq_fasta_format_error": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "bzq_fasta_error_open_record": (C.c_int64, [C.c_void_p]),
    "bzq_fasta_copy_to_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bzq_bgzf_scan": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(BzqBgzfBlock), C.c_int64, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bzq_bgzf_inflate": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(BzqBgzfBlock), C.c_int64, C.c_void_p, C.c_uint64]),
    "bzq_gzip_open": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bzq_gzip_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    "bzq_gzip_decode": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "bzq_gzip_stage": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "bzq_gzip_finished": (C.c_int32, [C.c_void_p]),
    "bzq_gzip_get_stats": (C.c_int32, [C.c_void_p, C.POINTER(BzqGzipStats)]),
    "bzq_gzip_last_error": (C.c_char_p, [C.c_void_p]),
    "bzq_gzip_close": (None, [C.c_void_p]),
    "bzq_fasta_plan_shards": (C.c_int32, [C.POINTER(BzqFastaShardSummary), C.c_int32, C.POINTER(BzqFastaShardPlan)]),
    "bzq_fasta_shard_scan": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(BzqFastaShardSummary)]),
    "bzq_fasta_shard_stitch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                           C.POINTER(BzqFastaShardResult)]),
    "bzq_fasta_ingest_open": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]),
    "bzq_fasta_ingest_next": (C.c_int32, [C.c_void_p, C.POINTER(BzqFastaChunk), C.POINTER(C.c_uint64)]),
    "bzq_fasta_ingest_get_stats": (C.c_int32, [C.c_void_p, C.POINTER(BzqIngestStats)]),
    "bzq_fasta_ingest_close": (None, [C.c_void_p]),
    "bzq_fasta_generate_synthetic_device": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                                        C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
}

_lib = None


class LibraryMissing(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """A PyTorch wheel bundles its own libamdhip64.so.7; the system ROCm has one with the same SONAME.  Whichever is
    loaded first serves both, and torch does not find its GPUs on the system copy.  So when torch is installed (and
    not imported yet) its copy is loaded first -- without importing torch -- and this library binds to it; device
    pointers and streams are then shared with torch in either import order."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # no torch, or an unusual layout: the system HIP runtime is used


def lib():
    """Load libblazeseq_hip.so.  Raises LibraryMissing (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  blazeseq_amd has no CPU fallback.")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
