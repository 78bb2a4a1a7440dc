"""-m gpu: the FIRST chunk of a FRESH ctx, and the re-make paths under it.

Round 5 saw one timing-dependent wrong answer (0 records / SEP_NO_PLUS instead of 6 726 / EOF on a chunk of tiny records in views
mode) and could not explain it.  Round 6 did: bzq_create zeroed the views pool's ticket with hipMemset -- on the NULL
stream, asynchronous to the host -- while the ctx stream is a non-blocking stream, so nothing ordered that fill in front of the first
chunk's pass A; a fill that landed while pass A was handing out pool slots reset the ticket under it, two tiles got the same slot and
one overwrote the other's line entries.  It needs a fresh ctx, a first chunk with pool tiles (records of a few bytes) and the fill
to be late -- which is why a campaign that keeps 40 ctxs alive met it once in 25 000 streams and a single-ctx stress never did.
The fill is on the ctx stream now.  Here: (1) the deterministic reproducer -- a spin kernel holds the NULL stream while the ctx is
created and its first chunk, 0.8 GB with pool tiles all along, is parsed -- rounds 4-5's create (kept behind BZQ_POOL_ZERO=0 as this
test's hook) gives wrong results whenever the spin ends inside pass A, the shipped create never; (2) bounded stress (8 s each) of fresh ctxs on
the two campaign streams that failed (views mode: pool tiles, record arrays that overflow and are re-made) and a batch-mode twin."""
import os
import time

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _campaign_stream(seed):
    """tests/fuzz_campaign.py's stream of that seed (kind "tiny")."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    from fastq_fuzz import rand_stream, rand_record
    src = open(os.path.join(here, "fuzz_campaign.py")).read()
    g = {"np": np, "rand_stream": rand_stream, "rand_record": rand_record}
    exec(compile(src[src.index("def make_stream(rng):"):src.index("ap = argparse.ArgumentParser()")], "fc", "exec"), g)
    data, kind = g["make_stream"](np.random.default_rng(seed))
    assert kind == "tiny"
    return np.frombuffer(bytes(data), dtype=np.uint8).copy()


def _pool_chunk(total_mb):
    """ordinary 150 bp reads with a stretch of 8-byte records (2048 per 16 KiB tile: a pool tile) every ~520 KB: ~6 % of the tiles"""
    normal = O.generate_synthetic(1600, 150, 150, 0, 40, "sanger").tobytes()
    tiny = b"@\nA\n+\n!\n" * 2600
    piece = normal + tiny
    reps = max(1, total_mb * 1_000_000 // len(piece))
    return np.frombuffer(piece * reps, dtype=np.uint8), reps * 4200


def test_a_late_fill_of_the_pool_ticket_is_the_round_5_mismatch_and_the_shipped_create_is_immune(monkeypatch):
    import torch
    import blazeseq_amd as B
    data, want = _pool_chunk(800)
    d_chunk = torch.from_numpy(data.copy()).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(100_000_000); torch.cuda.synchronize()
    rate = 100_000_000 / (time.perf_counter() - t0)          # the spin kernel's counter, ticks per second

    def run(spin_us):
        torch.cuda.synchronize()
        if spin_us:
            torch.cuda._sleep(int(spin_us * 1e-6 * rate))     # on torch's current stream = the NULL stream
        ctx = B.Context(B.ParserConfig(views_only=True), "generic", 4096, 0, min_record_bytes=8)
        ctx.submit_device(d_chunk.data_ptr(), data.size, 0, True)
        r = ctx.result()
        got = (int(r.n_records), int(r.status))
        ctx.close()
        return got

    spins = [0] + list(range(100, 2500, 50))                  # pass A of 0.8 GB takes ~0.15 ms and starts 0.5-0.7 ms after the spin was launched (create + submit on the host)
    monkeypatch.setenv("BZQ_POOL_ZERO", "1")
    assert run(0) == (want, 6)                                # (first-use costs out of the way)
    wrong_new = [s for s in spins for _ in range(2) if run(s) != (want, 6)]
    assert wrong_new == [], wrong_new                         # the shipped create: the fill is ordered in front of pass A -- THE regression assertion
    monkeypatch.setenv("BZQ_POOL_ZERO", "0")                  # rounds 4-5: hipMemset on the NULL stream (test hook)
    wrong_old = [s for s in spins for _ in range(2) if run(s) != (want, 6)]
    if not wrong_old:                                         # (a finer second sweep before giving up on this box's timing)
        wrong_old = [s for s in range(100, 2500, 20) if run(s) != (want, 6)]
    monkeypatch.setenv("BZQ_POOL_ZERO", "1")
    # the root cause shown, not assumed: with the fill held back into pass A the old create DOES give wrong answers (15 of 120 in
    # profiles/r6_pool_zero_race.log; 3 of 3 suite runs of round 6).  Where the window lies depends on the host's create + submit
    # time, so a box on which the sweep misses it is reported, not failed: the regression assertion is the one above.
    if not wrong_old:
        import warnings
        warnings.warn("the null-stream fill did not reproduce the round-5 mismatch on this box (timing window missed by the sweep)")


@pytest.mark.parametrize("seed", [2723, 2982])
def test_fresh_ctx_views_stress_on_the_campaign_streams_that_failed(seed):
    """8 s of: create a ctx, parse the stream once (pool tiles; the record arrays overflow and are re-made), compare, close --
    with garbage in freshly freed device memory.  Full output parity once, (records, status, consumed, newlines) every time."""
    import torch
    from gpu_util import make_pair, check_views_against_oracle
    data = _campaign_stream(seed)
    kw = dict(check_ascii=True, check_quality=True, quality_schema="sanger", views_only=True, buffer_capacity=64) if seed == 2723 else \
         dict(views_only=True, buffer_capacity=65536)
    bs = 100 if seed == 2723 else 7
    ctx, ocfg = make_pair(batch_size=bs, single_pass=False, **kw)
    res, f = check_views_against_oracle(ctx, ocfg, data, is_eof=True, what=f"seed {seed}")
    want = (f.n_records, f.term_code, f.consumed, f.n_newlines)
    ctx.close()
    junk = [torch.full((64,), 0x7F7F7F7F, dtype=torch.int32, device="cuda") for _ in range(256)]
    torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < 8:
        ctx, _ = make_pair(batch_size=bs, single_pass=False, **kw)
        r = ctx.parse(data, 0, True)
        got = (int(r.n_records), int(r.status), int(r.bytes_consumed), int(r.total_newlines))
        if got != want:
            bad.append((n, got))
        ctx.close()
        n += 1
    assert not bad and n > 1000, (n, bad[:5])


def test_fresh_ctx_batch_mode_stress_with_the_record_arrays_re_made():
    """The batch-mode twin: tiny records (every tile on the serial in-kernel path, rec_overflow -> re-size -> emit again) as the first
    chunk of a fresh ctx, 6 s; columns and ends compared by digest every time."""
    import hashlib
    import blazeseq_amd as B
    data = np.frombuffer(b"@\n\n+\n\n" * 9000 + b"@a\nC\n+\n!\n" * 3000 + O.generate_synthetic(500, 20, 90, 0, 40, "sanger").tobytes(), dtype=np.uint8)
    f = O.flat_parse(data, O.make_config(batch_size=256))
    dig = lambda *arrs: hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for a in arrs)).hexdigest()
    want = (f.n_records, f.term_code, dig(f.seq_bytes, f.qual_bytes, f.id_bytes, f.record_end))
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < 6:
        ctx = B.Context(B.ParserConfig(), "generic", 256, 0)
        r = ctx.parse(data, 0, True)
        got = (int(r.n_records), int(r.status), dig(r.seq(), r.qual(), r.id(), r.record_end()))
        if got != want:
            bad.append((n, got[:2]))
        ctx.close()
        n += 1
    assert not bad and n > 200, (n, bad[:5])
