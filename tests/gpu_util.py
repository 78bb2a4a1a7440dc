"""Helpers for the -m gpu parity tests: run the HIP path through the C ABI, compare with the oracle."""
import os

import numpy as np

import blazeseq_amd as B
from blazeseq_amd import _lib as L
from oracle import oracle as O


# The product library holds ONE batch-mode implementation (aggregate + scan + emit, "False" below).  The single-launch
# variants (True = look-back, "svc" = prefix-service workgroup, "hier" = two-level look-back) and the first-generation
# kernels ("v1") are independent implementations kept as cross-checks in an EXPERIMENTS build (libblazeseq_hip_exp.so):
# tests/test_gpu_experiments.py re-runs the parity files against that library with BZQ_TEST_EXPERIMENTS=1.
EXPERIMENTS = os.environ.get("BZQ_TEST_EXPERIMENTS", "0") == "1"
# "stream" = k_stream (bzq_stream.hpp): one read of the input, super-tiles staged in registers, ticket-ordered class-form
# look-back; falls back to the two-pass kernels when an id has leading / trailing spaces.  Also EXPERIMENTS only.
VARIANTS = [False] + ([True, "v1", "svc", "hier", "stream"] if EXPERIMENTS else [])
VARIANTS_LB = [False] + ([True, "stream"] if EXPERIMENTS else [])
SHARD_VARIANTS = [False] + ([True, 2, 3] if EXPERIMENTS else [])


def make_pair(batch_size=4096, schema="generic", pass_bytes=0, min_record_bytes=32, single_pass=False, **kw):
    """(Context, oracle config) with the same ParserConfig."""
    okw = {k: v for k, v in kw.items() if k not in ("compat_simd_width", "emit_offsets", "views_only")}
    cfg = B.ParserConfig(**kw)
    name = cfg.quality_schema if cfg.quality_schema else schema
    ctx = B.Context(cfg, schema, batch_size, 0, pass_bytes=pass_bytes, min_record_bytes=min_record_bytes)
    # single_pass: True = one fused launch with in-kernel look-back; False = aggregate + scan + emit
    # (table-driven v2 kernels); "v1" = the first-generation two-pass kernels
    if single_pass == "stream":
        ctx.set_option("stream", 1)
    elif single_pass is not False:
        ctx.set_option("single_pass", {"svc": 2, "hier": 3}.get(single_pass, int(single_pass is True)))
        ctx.set_option("kernels_v2", 0 if single_pass == "v1" else 1)
    okw.pop("quality_schema", None)
    ocfg = O.make_config(quality_schema=name, simd_width=kw.get("compat_simd_width", 0), batch_size=batch_size, **okw)
    return ctx, ocfg


def check_against_oracle(ctx, ocfg, data, is_eof=True, offsets=False, what=""):
    """Parse `data` on the GPU and with the flat oracle; every output must be bit-identical."""
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    res = ctx.parse(data, 0, is_eof)
    f = O.flat_parse(data, ocfg, is_eof=is_eof)
    tag = f"{what} n={data.size}"
    assert int(res.n_records) == f.n_records, (tag, int(res.n_records), f.n_records, res.status, f.term_code)
    assert res.status == f.term_code, (tag, res.status, f.term_code, ctx.format_error(), f.term_msg)
    if res.status != L.OK:
        assert ctx.format_error() == f.term_msg, (tag, ctx.format_error(), f.term_msg)
        assert (int(res.error_record) if res.status != L.EOF else -1) == f.term_record, tag
    assert int(res.bytes_consumed) == f.consumed, (tag, int(res.bytes_consumed), f.consumed)
    assert int(res.total_newlines) == f.n_newlines, tag
    n = f.n_records
    if n:
        np.testing.assert_array_equal(res.ends(), f.ends, err_msg=tag + " ends")
        np.testing.assert_array_equal(res.id_ends(), f.id_ends, err_msg=tag + " id_ends")
        np.testing.assert_array_equal(res.record_end(), f.record_end, err_msg=tag + " record_end")
        bs = ocfg.batch_size
        r = np.arange(n)
        b0 = (r // bs) * bs
        base_e = np.where(b0 > 0, f.ends[np.maximum(b0 - 1, 0)], 0)
        base_i = np.where(b0 > 0, f.id_ends[np.maximum(b0 - 1, 0)], 0)
        np.testing.assert_array_equal(res.batch_ends(), f.ends - base_e, err_msg=tag + " batch_ends")
        np.testing.assert_array_equal(res.batch_id_ends(), f.id_ends - base_i, err_msg=tag + " batch_id_ends")
        assert int(res.qual_bytes) == f.qual_bytes.size and int(res.id_bytes) == f.id_bytes.size, tag
        assert int(res.seq_bytes) == f.seq_bytes.size, (tag, int(res.seq_bytes), f.seq_bytes.size)
        np.testing.assert_array_equal(res.seq(), f.seq_bytes, err_msg=tag + " seq column")
        np.testing.assert_array_equal(res.qual(), f.qual_bytes, err_msg=tag + " qual column")
        np.testing.assert_array_equal(res.id(), f.id_bytes, err_msg=tag + " id column")
        if offsets:
            np.testing.assert_array_equal(res.header_start(), f.header_start, err_msg=tag + " header_start")
            np.testing.assert_array_equal(res.seq_start(), f.seq_start, err_msg=tag + " seq_start")
            np.testing.assert_array_equal(res.sep_start(), f.sep_start, err_msg=tag + " sep_start")
            np.testing.assert_array_equal(res.qual_start(), f.qual_start, err_msg=tag + " qual_start")
    return res, f


def check_views_against_oracle(ctx, ocfg, data, is_eof=True, what=""):
    """Views mode (config.views_only): RecordOffsets + id spans into the chunk must reproduce the oracle's records:
    same count, terminal event, error text, offsets; id / sequence / quality spans must hold the oracle's bytes."""
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    res = ctx.parse(data, 0, is_eof)
    f = O.flat_parse(data, ocfg, is_eof=is_eof)
    tag = f"views {what} n={data.size}"
    assert int(res.n_records) == f.n_records, (tag, int(res.n_records), f.n_records, res.status, f.term_code)
    assert res.status == f.term_code, (tag, res.status, f.term_code, ctx.format_error(), f.term_msg)
    if res.status != L.OK:
        assert ctx.format_error() == f.term_msg, (tag, ctx.format_error(), f.term_msg)
        assert (int(res.error_record) if res.status != L.EOF else -1) == f.term_record, tag
    assert int(res.bytes_consumed) == f.consumed, (tag, int(res.bytes_consumed), f.consumed)
    assert int(res.total_newlines) == f.n_newlines, tag
    n = f.n_records
    if n:
        np.testing.assert_array_equal(res.record_end(), f.record_end, err_msg=tag + " record_end")
        np.testing.assert_array_equal(res.header_start(), f.header_start, err_msg=tag + " header_start")
        np.testing.assert_array_equal(res.seq_start(), f.seq_start, err_msg=tag + " seq_start")
        np.testing.assert_array_equal(res.sep_start(), f.sep_start, err_msg=tag + " sep_start")
        np.testing.assert_array_equal(res.qual_start(), f.qual_start, err_msg=tag + " qual_start")
        ids, idl = res.id_start(), res.id_len()
        np.testing.assert_array_equal(np.cumsum(idl.astype(np.int64)), f.id_ends, err_msg=tag + " id lengths")
        i0 = np.concatenate([[0], f.id_ends[:-1]])
        for r in range(n) if n <= 3000 else list(range(0, n, max(1, n // 3000))):
            a, b = int(ids[r]), int(ids[r]) + int(idl[r])
            assert data[a:b].tobytes() == f.id_bytes[int(i0[r]):int(f.id_ends[r])].tobytes(), (tag, "id bytes", r)
    return res, f
