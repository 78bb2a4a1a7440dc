"""Round 4: the emit kernel writes the per-batch ends and does the record-length check itself (FusedArgs::fold: batch bases from
k_batch_bases between the scan and the emit, no k_rebase pass over the per-record arrays).  Everything the chunk delivers must
stay bit-identical to the oracle -- with the fold, and with option fold_rebase = 0 (the k_rebase path, still used for batch sizes
below 256, sub-chunk passes and the SIMD-width quirk) -- on streams with MANY batch boundaries, records that straddle tiles,
records refused by the reference's buffer limit (parser.mojo:484-492), tiles of tiny records (the serial in-kernel path) and
shards with head lines.  Also: the exact pass A is sticky after a contradicted hypothesis (a CRLF file is not parsed twice)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import blazeseq_amd as B
from blazeseq_amd import _lib as L
from oracle import oracle as O
from fastq_fuzz import rand_stream
from gpu_util import make_pair, check_against_oracle


def folded(ctx):
    return L.lib().bzq_set_option(ctx.h, b"last_folded", 0) == 1


@pytest.mark.parametrize("seed", range(10))
def test_fold_equals_oracle_and_rebase(seed):
    rng = np.random.default_rng(4000 + seed)
    max_len = int(rng.choice([3, 30, 150, 150, 400, 5000, 40000]))
    nrec = int(rng.integers(3000, 30000)) if max_len <= 400 else int(rng.integers(300, 1500))
    data = rand_stream(rng, n_records=nrec, max_len=max_len, dirty=float(rng.choice([0.0, 0.0, 0.0005])),
                       crlf=bool(rng.random() < 0.2))
    bs = int(rng.choice([256, 257, 300, 1000, 4096]))
    cap = int(rng.choice([max(32, max_len), 2 * max_len + 40, 65536]))
    for kw in (dict(), dict(check_ascii=True, check_quality=True), dict(buffer_capacity=cap),
               dict(buffer_capacity=cap, buffer_growth_enabled=True, buffer_max_capacity=4 * cap)):
        for fold in (1, 0):
            ctx, oc = make_pair(batch_size=bs, emit_offsets=bool(seed & 1), **kw)
            ctx.set_option("fold_rebase", fold)
            check_against_oracle(ctx, oc, data, offsets=bool(seed & 1), what=f"fold={fold} seed{seed} bs={bs} {kw}")
            assert folded(ctx) == bool(fold), (seed, bs, kw)
            if fold and seed % 3 == 0:
                ctx.set_option("force_dense", 1)   # every tile through the serial in-kernel path
                check_against_oracle(ctx, oc, data, offsets=bool(seed & 1), what=f"fold dense seed{seed} bs={bs} {kw}")
            ctx.close()


def test_fold_is_off_where_it_cannot_be_used():
    data = rand_stream(np.random.default_rng(7), n_records=2000, max_len=100, dirty=0.0, tail=0)
    for kw, bs, pb, want in ((dict(), 4096, 0, True), (dict(), 255, 0, False), (dict(), 4096, 16 * 1024, False),
                             (dict(check_quality=True, compat_simd_width=32), 4096, 0, False),
                             (dict(check_quality=True), 4096, 0, True)):
        ctx, oc = make_pair(batch_size=bs, pass_bytes=pb, **kw)
        check_against_oracle(ctx, oc, data, what=f"{kw} bs={bs} pb={pb}")
        assert folded(ctx) == want, (kw, bs, pb)
        ctx.close()


@pytest.mark.parametrize("long_at", [0, 1, 255, 256, 700, 1999])
@pytest.mark.parametrize("long_len", [20000, 70000])
def test_a_record_the_buffer_cannot_hold_is_refused_where_the_reference_refuses_it(long_at, long_len):
    """One record of 2 x long_len + ~12 bytes among 2000 short ones: it spans 3-9 tiles, so the tile its quality line ends in
    learns where it started from tileP / tile_last (prev_record_end) -- or gives up on a walk longer than the limit."""
    recs = [b"@r%d\nACGT\n+\nIIII\n" % i for i in range(2000)]
    recs[long_at] = b"@long\n" + b"A" * long_len + b"\n+\n" + b"I" * long_len + b"\n"
    data = b"".join(recs)
    for cap, growth in ((65536, False), (30000, False), (2 * long_len + 12, False), (2 * long_len + 11, False), (16384, True), (256, False)):
        kw = dict(buffer_capacity=cap)
        if growth:
            kw.update(buffer_growth_enabled=True, buffer_max_capacity=4 * cap)
        for fold in (1, 0):
            ctx, oc = make_pair(batch_size=256, **kw)
            ctx.set_option("fold_rebase", fold)
            res, f = check_against_oracle(ctx, oc, data, what=f"long@{long_at} len={long_len} cap={cap} growth={growth} fold={fold}")
            ctx.close()


def test_tiny_records_take_the_serial_path_with_many_boundaries_per_tile():
    data = b"@\n\n+\n\n" * 40000 + b"@a\nA\n+\nI\n" * 5000   # 2730 / 1638 records per 16 KiB tile
    for bs in (256, 1000):
        ctx, oc = make_pair(batch_size=bs, min_record_bytes=6)
        check_against_oracle(ctx, oc, data, what=f"tiny bs={bs}")
        assert folded(ctx)
        ctx.close()
    ctx, oc = make_pair(batch_size=256)   # the sizing hint is too optimistic: the chunk is re-run after an exact re-size
    check_against_oracle(ctx, oc, data, what="tiny, re-sized")
    ctx.close()


def test_full_batches_of_150bp_reads_and_the_host_boundary_table():
    """The bench shape at 1/50 scale: batch views must come out of the host table without touching the device."""
    n = 200_000
    data = O.generate_synthetic(n, 150, 150, 33, 73, "generic")
    ctx, oc = make_pair(batch_size=4096)
    res, f = check_against_oracle(ctx, oc, data, what="150bp")
    assert folded(ctx)
    nb = (n + 4095) // 4096
    views, got = ctx.batches(4096)
    assert got == nb
    views = [views[k] for k in range(nb)]
    e_prev = 0
    for k, v in enumerate(views):
        lo, hi = k * 4096, min(n, (k + 1) * 4096)
        assert int(v.num_records) == hi - lo
        assert int(v.seq_len) == int(f.ends[hi - 1]) - e_prev
        e_prev = int(f.ends[hi - 1])
    ctx.close()


def test_crlf_stream_is_not_parsed_twice_per_chunk():
    """example_dos.fastq's property at scale: every id loses its '\\r' to _strip_spaces, so pass A's hypothesis fails on EVERY
    chunk.  The first contradiction makes the exact pass A sticky."""
    rng = np.random.default_rng(11)
    data = rand_stream(rng, n_records=60000, max_len=100, dirty=0.0, crlf=True, tail=0)
    ocfg = O.make_config(batch_size=4096)
    want = O.flat_parse(np.frombuffer(data, dtype=np.uint8), ocfg, is_eof=True)
    p = B.FastqParser(data, batch_size=4096, chunk_bytes=64 * 1024)
    n = 0
    for b in p.batches():
        n += b.num_records()
    assert n == want.n_records
    fb = L.lib().bzq_set_option(p._ctx.h, b"stream_fallbacks", 0)
    chunks = L.lib().bzq_set_option(p._ctx.h, b"n_submits", 0)
    assert chunks >= 20, chunks
    assert 1 <= fb <= 3 and fb <= chunks // 8, (fb, chunks)   # 16 sticky chunks, then 32, 64 ...: 3 retries cover 1 + 16 + 1 + 32 + 1 + 64 chunks
    # and with the stickiness off every chunk is repeated (what round 3 did)
    p2 = B.FastqParser(data, batch_size=4096, chunk_bytes=64 * 1024)
    p2._ctx.set_option("pass_a_sticky", 0)
    n2 = sum(b.num_records() for b in p2.batches())
    assert n2 == n
    assert L.lib().bzq_set_option(p2._ctx.h, b"stream_fallbacks", 0) >= chunks - 2
