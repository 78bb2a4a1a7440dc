"""-m gpu: views mode (ParserConfig(views_only=True) -> bzq_config.views_only): the device analogue of parser.views()
(parser.mojo:253-258).  Same oracle, same corpus, same fuzz generators as the batch path; the outputs compared are the
RecordOffsets columns and the stripped-id spans."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from fastq_fuzz import rand_stream
from gpu_util import make_pair, check_views_against_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
CORPUS = json.load(open(os.path.join(HERE, "golden", "corpus_expected.json")))


@pytest.mark.parametrize("bytes_path", [0, 1])
def test_views_inline_known_answers(bytes_path):
    ctx, oc = make_pair(views_only=True, single_pass=False)
    ctx.set_option("views_bytes", bytes_path)   # 0: line entries left by pass A (one read of the input); 1: two reads
    for data in (b"@r1\nACGT\n+\n!!!!\n@r2\nTGCA\n+\n####\n", b"", b"\n", b"@", b"@a\nA\n+\n!\n",
                 b"r1\nACGT\n+\n!!!!\n", b"@r1\nACGT\n+\n!!!\n", b"@r1\nACGT\n-\n!!!!\n",
                 b"@a\nAC\n+\n!!\n@b\nACGT\n+\n!!", b"@a\nAC\n+\n!!\n@b\nAC\n+\n \t", b"@a\nAC\n+\n!!\n\n",
                 b"@only\nACGT", b"@a\nAC\n+\n!!\n@only\nACGT", b"@id\r\nACGT\r\n+\r\n!!!!\r\n",
                 b"@ \t id with spaces \t\nAC\n+\n!!\n@\nAC\n+\n!!\n@   \nAC\n+\n!!\n", b"\n\n\n\n", b"\n\n\n\n\n\n\n\n\n"):
        check_views_against_oracle(ctx, oc, data, what=repr(data[:20]))


@pytest.mark.parametrize("cfgname", ["default", "validated_generic", "validated_schema", "validated_schema_simd32", "cap64", "cap64_growth"])
def test_views_corpus(cfgname, corpus_dir):
    for name, e in sorted(CORPUS.items()):
        data = open(os.path.join(corpus_dir, name), "rb").read()
        sc = e["schema"]
        kw = {"default": {}, "validated_generic": dict(check_ascii=True, check_quality=True),
              "validated_schema": dict(check_ascii=True, check_quality=True, quality_schema=sc),
              "validated_schema_simd32": dict(check_ascii=True, check_quality=True, quality_schema=sc, compat_simd_width=32),
              "cap64": dict(buffer_capacity=64),
              "cap64_growth": dict(buffer_capacity=64, buffer_growth_enabled=True, buffer_max_capacity=1 << 20)}[cfgname]
        ctx, oc = make_pair(views_only=True, single_pass=False, **kw)
        if cfgname in ("cap64",):
            ctx.set_option("views_bytes", 1)
        res, f = check_views_against_oracle(ctx, oc, data, what=f"{name}/{cfgname}")
        g = e[cfgname]
        assert (int(res.n_records), res.status, ctx.format_error().decode("latin-1") if res.status else "") == \
               (g["n_records"], g["term_code"], g["term_msg"]), name
        ctx.close()


@pytest.mark.parametrize("seed", range(12))
def test_views_fuzz(seed):
    rng = np.random.default_rng(7000 + seed)
    for rep in range(5):
        big = rep == 4
        data = rand_stream(rng, n_records=int(rng.integers(0, 60)) if not big else int(rng.integers(300, 3000)),
                           max_len=int(rng.integers(1, 80)) if not big else int(rng.choice([30, 150, 5000, 40000]) if seed % 2 else 150),
                           dirty=float(rng.choice([0.0, 0.02, 0.1])) if not big else float(rng.choice([0, 0.001])),
                           crlf=bool(rng.random() < 0.15))
        for kw in (dict(), dict(check_ascii=True, check_quality=True),
                   dict(check_ascii=True, check_quality=True, quality_schema="solexa", compat_simd_width=16),
                   dict(buffer_capacity=48), dict(buffer_capacity=48, buffer_growth_enabled=True, buffer_max_capacity=200)):
            ctx, oc = make_pair(views_only=True, single_pass=False, batch_size=int(rng.choice([1, 3, 4096])),
                                pass_bytes=int(rng.choice([0, 0, 16 * 1024, 64 * 1024])) if not big else 0, **kw)
            ctx.set_option("views_bytes", int(rng.random() < 0.3))
            check_views_against_oracle(ctx, oc, data, what=f"seed{seed}/{rep}/{kw}")
            check_views_against_oracle(ctx, oc, data, is_eof=False, what=f"chunk seed{seed}/{rep}/{kw}")
            if rep in (0, 4):
                ctx.set_option("force_dense", 1)
                check_views_against_oracle(ctx, oc, data, what=f"dense seed{seed}/{kw}")
            ctx.close()


def test_views_long_space_runs_saturate_the_line_entries():
    """Id space runs of 255+ bytes do not fit a line entry: the span is recomputed from the bytes."""
    recs = []
    for i, (lead, trail) in enumerate([(0, 0), (254, 3), (255, 0), (0, 255), (300, 700), (5000, 1), (20000, 20000), (256, 256)]):
        recs.append(b"@" + b" " * lead + (b"id%d" % i if i != 6 else b"") + b"\t" * trail + b"\nACGT\n+\nIIII\n")
    data = b"".join(recs)
    for bp in (0, 1):
        ctx, oc = make_pair(views_only=True, single_pass=False)
        ctx.set_option("views_bytes", bp)
        res, f = check_views_against_oracle(ctx, oc, data, what="space runs")
        assert f.n_records == 8
        ctx.close()


def test_views_seams_between_listed_and_entry_tiles():
    """Found by tests/fuzz_campaign.py --views (seed 37): a tile with more newlines than line entries fit goes through
    the byte-level kernel, its neighbours through the entries; lines (and header ids) that straddle such a seam, or
    start exactly on it, must be written by exactly one of them."""
    rec = b"@r\nACGT\n+\nIIII\n"                      # 15 bytes, 4 newlines: ~4400 newlines per tile -> listed
    sparse = b"@read with a long id %d\n" + b"ACGT" * 40 + b"\n+\n" + b"I" * 160 + b"\n"
    for role in range(4):
        ends = [2, 7, 9, 14]
        body = rec * (16384 // 15 - 2)
        need = 16383 - (len(body) + ends[role])
        dense = b"@" + b"x" * (need - 14) + b"\nACGT\n+\nIIII\n" + body + rec   # newline of line `role` of the last rec on byte 16383
        data = dense + b"".join(sparse % i for i in range(200)) + rec * 3000 + b"".join(sparse % i for i in range(100))
        d = np.frombuffer(data, dtype=np.uint8)
        assert d[16383] == 10
        for bp in (0, 1):
            ctx, oc = make_pair(views_only=True, single_pass=False)
            ctx.set_option("views_bytes", bp)
            check_views_against_oracle(ctx, oc, d, what=f"seam role {role}")
            ctx.close()
    # header lines that start in a listed tile and end in an entry tile, and the reverse, with space runs in the id
    for shift in range(0, 40, 3):
        head = rec * ((16384 - 20 - shift) // 15)
        hdr = b"@" + b" " * 7 + b"id across the seam" + b"\t" * 9 + b"\nACGT\n+\nIIII\n"
        data = head + hdr + b"".join(sparse % i for i in range(120)) + hdr * 3 + rec * 1200
        for bp in (0, 1):
            ctx, oc = make_pair(views_only=True, single_pass=False)
            ctx.set_option("views_bytes", bp)
            check_views_against_oracle(ctx, oc, np.frombuffer(data, dtype=np.uint8), what=f"seam header {shift}")
            ctx.close()


def test_views_api_and_spans():
    """bzq_views: zero-copy spans of a record range; the spans hold the record's bytes."""
    import ctypes as C
    import torch
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    data = O.generate_synthetic(5000, 30, 200, 0, 40, "sanger")
    f = O.flat_parse(data, O.make_config())
    ctx = B.Context(B.ParserConfig(views_only=True), "generic", 4096, 0)
    t = torch.from_numpy(data.copy()).cuda()
    ctx.submit_device(t.data_ptr(), t.numel(), 0, True)
    res = ctx.result()
    assert int(res.n_records) == 5000 and res.status == 6 and not res.d_seq and not res.d_ends
    v = L.BzqDeviceViews()
    assert L.lib().bzq_views(ctx.h, 1000, 64, C.byref(v)) == 0
    assert v.num_records == 64 and v.chunk == t.data_ptr() and v.first_record == 1000
    ss, se, qs, re_ = (np.empty(64, dtype=np.int64) for _ in range(4))
    for arr, ptr in ((ss, v.seq_start), (se, v.sep_start), (qs, v.qual_start), (re_, v.record_end)):
        ctx.copy_to_host(arr, ptr, 64 * 8)
    e0 = np.concatenate([[0], f.ends])
    for k in range(64):
        r = 1000 + k
        assert data[ss[k]:se[k] - 1].tobytes() == f.seq_bytes[e0[r]:e0[r + 1]].tobytes()
        assert data[qs[k]:re_[k]].tobytes() == f.qual_bytes[e0[r]:e0[r + 1]].tobytes()
    b = L.BzqDeviceBatch()
    assert L.lib().bzq_batch_view(ctx.h, 0, 10, C.byref(b)) < 0      # no columns in this mode
    ctx.close()


@pytest.mark.parametrize("seed", range(4))
def test_parser_views_iteration_matches_the_streaming_oracle(seed, capsys):
    """FastqParser(config=ParserConfig(views_only=True)).views() / next_view(): zero-copy spans of the host chunk at the
    offsets the views-mode kernels found, record by record like the reference's view iterator (parser.mojo:159-170,
    253-258, 628-661) -- same views, same terminal error as the streaming oracle."""
    import blazeseq_amd as B
    rng = np.random.default_rng(4200 + seed)
    data = rand_stream(rng, n_records=int(rng.integers(200, 2000)), max_len=150, dirty=[0.0, 0.0, 0.002, 0.0][seed], tail=[0, 3, 0, 1][seed])
    check = bool(seed & 1)
    ocfg = O.make_config(check_ascii=check, check_quality=check)
    sp = O.StreamParser(np.frombuffer(data, dtype=np.uint8), ocfg)
    want, werr = [], None
    while True:
        try:
            v = sp.next_view()
        except O.OracleError as e:
            werr = e
            break
        want.append((v.id, v.seq, v.qual))
    for chunk in (1 << 16, 1 << 26):
        p = B.FastqParser(data, config=B.ParserConfig(check_ascii=check, check_quality=check, views_only=True), chunk_bytes=chunk)
        got, gerr = [], None
        while True:
            try:
                v = p.next_view()
            except B.ParseError as e:
                gerr = e
                break
            got.append((bytes(v.id), bytes(v.sequence), bytes(v.quality)))
            assert len(v) == len(v.sequence) and v.byte_len() == 1 + len(v.id) + len(v.sequence) + len(v.quality) + 5
        assert got == want, (len(got), len(want))
        assert gerr.code == werr.code
        if werr.code != O.EOF:
            assert gerr.message.decode("latin-1") == str(werr)
        # the iterator form ends on any error (printing it unless it is EOF)
        p = B.FastqParser(data, config=B.ParserConfig(check_ascii=check, check_quality=check, views_only=True), chunk_bytes=chunk)
        assert sum(1 for _ in p.views()) == len(want)
        capsys.readouterr()
    # without views_only the same call is plumbing over batches
    assert [(bytes(v.id), bytes(v.sequence), bytes(v.quality)) for v in B.FastqParser(data, config=B.ParserConfig(check_ascii=check, check_quality=check)).views()] == want
    capsys.readouterr()
