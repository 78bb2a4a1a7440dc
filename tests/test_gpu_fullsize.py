"""-m gpu: BASELINE.json's FULL sizes, through size-independent properties that need no oracle run:

* config 2/3 (10 M x 150 bp, fixed record size): the three columns must equal slices of the input viewed as a
  [records, 318] matrix -- an exact, full-size check of every output byte; ends / id_ends / record_end are arithmetic
  progressions; the negative variant of config 3 (byte flips at records 0, 4095, 4096, R-1) must stop at exactly that
  record with exactly that code.
* config 4 (300 k long reads, variable record size): a checksum of checksums -- the byte histogram of the input equals
  the histograms of the three columns plus the structural bytes (one '@', one '+', four newlines per record) -- and the
  sequence / quality lengths follow the generator's length law.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
R2 = 10_000_000


def _gen(ctx, reads, read_len, lo, hi, schema, max_len=None):
    import torch
    nb = ctx.generate_synthetic_device(reads, read_len, lo, hi, schema, max_len=max_len)
    t = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
    ctx.generate_synthetic_device(reads, read_len, lo, hi, schema, t.data_ptr(), t.numel(), max_len=max_len)
    return t, nb


def _view(ptr, nbytes, dtype):
    """torch view of device memory owned by the ctx (no copy)."""
    import torch

    class _Ext:
        pass
    e = _Ext()
    item = {torch.uint8: "|u1", torch.int64: "<i8"}[dtype]
    size = nbytes if dtype == torch.uint8 else nbytes // 8
    e.__cuda_array_interface__ = {"shape": (size,), "typestr": item, "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(e, device="cuda")


@pytest.mark.parametrize("validate", [False, True])
def test_config2_and_3_full_size_columns_are_slices_of_the_input(validate):
    import torch
    import blazeseq_amd as B
    cfg = B.ParserConfig(check_ascii=validate, check_quality=validate, quality_schema="sanger" if validate else None)
    ctx = B.Context(cfg, "generic", 4096, 0)
    t, n = _gen(ctx, R2, 150, 33, 73, "generic")
    assert n == 318 * R2
    ctx.submit_device(t.data_ptr(), n, 0, True)
    res = ctx.result()
    assert int(res.n_records) == R2 and res.status == 6 and int(res.bytes_consumed) == n
    assert int(res.seq_bytes) == int(res.qual_bytes) == 150 * R2 and int(res.id_bytes) == 12 * R2
    m = t[:n].view(R2, 318)   # "@read_0000000\n" (14) + 150 + "\n+\n" (3) + 150 + "\n"
    assert torch.equal(_view(res.d_seq, 150 * R2, torch.uint8).view(R2, 150), m[:, 14:164])
    assert torch.equal(_view(res.d_qual, 150 * R2, torch.uint8).view(R2, 150), m[:, 167:317])
    assert torch.equal(_view(res.d_id, 12 * R2, torch.uint8).view(R2, 12), m[:, 1:13])
    r = torch.arange(1, R2 + 1, dtype=torch.int64, device="cuda")
    res._cumulative()   # (ABI 2: the chunk-cumulative arrays are derived on demand)
    assert torch.equal(_view(res.d_ends, 8 * R2, torch.int64), 150 * r)
    assert torch.equal(_view(res.d_id_ends, 8 * R2, torch.int64), 12 * r)
    assert torch.equal(_view(res.d_record_end, 8 * R2, torch.int64), 318 * r - 1)
    in_batch = (torch.arange(R2, dtype=torch.int64, device="cuda") % 4096) + 1
    assert torch.equal(_view(res.d_batch_ends, 8 * R2, torch.int64), 150 * in_batch)
    assert torch.equal(_view(res.d_batch_id_ends, 8 * R2, torch.int64), 12 * in_batch)
    del r, in_batch
    if validate:
        # negative variant: one 0x80 in a sequence, one 0x1F in a quality, at fixed records
        for rec, off, val, code in ((R2 - 1, 14 + 70, 0x80, 4), (4096, 167 + 3, 0x1F, 5), (4095, 14 + 149, 0x80, 4), (0, 167 + 149, 0x1F, 5)):
            pos = rec * 318 + off
            keep = int(t[pos].item())
            t[pos] = val
            ctx.submit_device(t.data_ptr(), n, 0, True)
            bad = ctx.result()
            assert bad.status == code and int(bad.n_records) == rec and int(bad.error_record) == rec, (rec, bad.status, int(bad.n_records))
            assert (b"Record number: %d" % (rec + 1)) in ctx.format_error()
            t[pos] = keep
    ctx.close()


def test_config4_full_size_checksum_of_checksums():
    import torch
    import blazeseq_amd as B
    reads = 300_000
    ctx = B.Context(B.ParserConfig(buffer_capacity=64 * 1024, check_ascii=True, check_quality=True, quality_schema="sanger"),
                    "generic", 4096, 0)
    t, n = _gen(ctx, reads, 200, 5, 30, "sanger", max_len=19_800)
    ctx.submit_device(t.data_ptr(), n, 0, True)
    res = ctx.result()
    assert int(res.n_records) == reads and res.status == 6 and int(res.bytes_consumed) == n
    # lengths follow utils.mojo:753-757
    i = torch.arange(reads, dtype=torch.int64, device="cuda")
    lens = 200 + (i * 31 + 7) % 19_601
    ends = torch.cumsum(lens, 0)
    res._cumulative()
    assert torch.equal(_view(res.d_ends, 8 * reads, torch.int64), ends)
    assert int(res.seq_bytes) == int(res.qual_bytes) == int(ends[-1].item()) and int(res.id_bytes) == 11 * reads
    rec_end = torch.cumsum(2 * lens + 17, 0) - 1          # "@read_000000\n" 13 + L + "\n+\n" 3 + L + "\n" 1 = 2L + 17
    assert torch.equal(_view(res.d_record_end, 8 * reads, torch.int64), rec_end)
    # checksum of checksums: histogram(input) == histogram(seq) + histogram(qual) + histogram(id) + structure
    whole = B.DeviceFastqBatch(ctx, ctx.batch_view(0, reads))
    h_in = np.zeros(256, dtype=np.uint64)
    from blazeseq_amd import _lib as L
    import ctypes as C
    out = (C.c_uint64 * 256)()
    assert L.lib().bzq_column_histogram(ctx.h, C.c_void_p(t.data_ptr()), n, out) == 0
    h_in = np.frombuffer(out, dtype=np.uint64).copy()
    assert L.lib().bzq_column_histogram(ctx.h, C.c_void_p(whole.id_buffer), int(res.id_bytes), out) == 0
    h_id = np.frombuffer(out, dtype=np.uint64).copy()
    h = whole.histogram("sequence") + whole.histogram("quality") + h_id
    h[ord("\n")] += 4 * reads; h[ord("@")] += reads; h[ord("+")] += reads
    np.testing.assert_array_equal(h, h_in)
    ctx.close()


@pytest.mark.parametrize("validate", [False, True])
def test_config2_and_3_full_size_views_mode(validate):
    """The same 10 M records through views mode (offsets + id spans, no columns; with validation: the line entries carry
    the two validation flags and the input is still read once): every offset is an arithmetic progression, and the
    config-3 byte flips stop at exactly that record with exactly that code."""
    import torch
    import blazeseq_amd as B
    cfg = B.ParserConfig(check_ascii=validate, check_quality=validate, quality_schema="sanger" if validate else None, views_only=True)
    ctx = B.Context(cfg, "generic", 4096, 0)
    t, n = _gen(ctx, R2, 150, 33, 73, "generic")
    ctx.submit_device(t.data_ptr(), n, 0, True)
    res = ctx.result()
    assert int(res.n_records) == R2 and res.status == 6 and int(res.bytes_consumed) == n
    r = torch.arange(R2, dtype=torch.int64, device="cuda")
    for ptr, off in ((res.d_header_start, 0), (res.d_seq_start, 14), (res.d_sep_start, 165), (res.d_qual_start, 167), (res.d_record_end, 317),
                     (res.d_id_start, 1)):
        assert torch.equal(_view(ptr, 8 * R2, torch.int64), 318 * r + off)
    del r
    if validate:
        for rec, off, val, code in ((R2 - 1, 14 + 70, 0x80, 4), (4096, 167 + 3, 0x1F, 5), (4095, 14 + 149, 0x80, 4), (0, 167 + 149, 0x1F, 5),
                                    (5_000_000, 3, 0x80, 4), (7, 167, 0x7F, 5)):
            pos = rec * 318 + off
            keep = int(t[pos].item())
            t[pos] = val
            ctx.submit_device(t.data_ptr(), n, 0, True)
            bad = ctx.result()
            assert bad.status == code and int(bad.n_records) == rec and int(bad.error_record) == rec, (rec, bad.status, int(bad.n_records))
            t[pos] = keep
    ctx.close()
