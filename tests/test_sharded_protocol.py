"""Multi-GPU shard stitch, protocol level (no GPU): plan_shards/head_bytes_of are pure functions of
the gathered shard summaries; with the oracle standing in for the device parse, shards cut at
arbitrary byte positions must reproduce the whole-stream parse.  The 2-rank gloo test runs the real
torch.distributed exchange (all_gather, isend/irecv, all_reduce) on CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blazeseq_amd import sharded
from oracle import oracle as O
from fastq_fuzz import rand_stream


def summary_of(buf: np.ndarray):
    nl = np.flatnonzero(buf == 10)
    first = [int(nl[i]) if i < nl.size else -1 for i in range(4)]
    return [int(buf.size), int(nl.size), *first, int(buf[0]) if buf.size else 10, int(buf[-1]) if buf.size else 10]


def parse_shard(buf_with_halo: np.ndarray, plan, is_last, cfg):
    """What bzq_submit_shard delivers, restated with the oracle: records whose header starts in the
    shard = a flat parse of [head_bytes, n + halo)."""
    region = buf_with_halo[plan.head_bytes:]
    return O.flat_parse(region, cfg, is_eof=is_last)


def stitch_in_process(data: np.ndarray, cuts, cfg):
    bounds = [0, *cuts, data.size]
    shards = [data[bounds[i]:bounds[i + 1]] for i in range(len(bounds) - 1)]
    plans = sharded.plan_shards([summary_of(s) for s in shards])
    ids, seqs, total = [], [], 0
    for r, (s, p) in enumerate(zip(shards, plans)):
        halo = shards[p.halo_src][:p.halo_bytes] if p.halo_src >= 0 else np.zeros(0, np.uint8)
        assert p.halo_bytes == halo.size
        is_last = all(x.size == 0 for x in shards[r + 1:])
        if s.size == 0:
            continue
        f = parse_shard(np.concatenate([s, halo]), p, is_last, cfg)
        assert f.term_code in (O.OK, O.EOF), (r, f.term_code, f.term_msg)
        assert p.lines_before == int(np.count_nonzero(data[:bounds[r]] == 10))
        total += f.n_records
        ids.append(f.id_bytes); seqs.append(f.seq_bytes)
    return total, np.concatenate(ids) if ids else np.zeros(0, np.uint8), np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)


@pytest.mark.parametrize("seed", range(30))
def test_arbitrary_cuts_reproduce_whole_parse(seed):
    rng = np.random.default_rng(seed)
    data = np.frombuffer(rand_stream(rng, n_records=int(rng.integers(40, 200)), max_len=int(rng.integers(1, 80)),
                                     dirty=0.0, tail=0, crlf=bool(rng.random() < 0.2)), dtype=np.uint8)
    cfg = O.make_config()
    whole = O.flat_parse(data, cfg)
    assert whole.term_code == O.EOF
    for P in (2, 3, 5, 8):
        rec = data.size / max(1, whole.n_records)
        lo = int(3 * rec) + 200
        if data.size < P * lo:
            continue
        # every shard longer than the longest record (max_len*2+30), otherwise arbitrary byte cuts
        cuts = sorted(int(x) for x in rng.integers(lo, data.size - lo, P - 1))
        cuts = [c for i, c in enumerate(cuts) if i == 0 or c - cuts[i - 1] > lo]
        total, ids, seqs = stitch_in_process(data, cuts, cfg)
        assert total == whole.n_records
        np.testing.assert_array_equal(ids, whole.id_bytes)
        np.testing.assert_array_equal(seqs, whole.seq_bytes)


def test_cut_positions_exhaustive_small():
    data = np.frombuffer(b"@r1\nACGT\n+\n!!!!\n@r2 x\nTG\n+r2\n##\n@r3\nA\n+\n!\n" * 3, dtype=np.uint8)
    cfg = O.make_config()
    whole = O.flat_parse(data, cfg)
    for c in range(20, data.size - 20):
        total, ids, seqs = stitch_in_process(data, [c], cfg)
        assert total == whole.n_records, c
        np.testing.assert_array_equal(ids, whole.id_bytes)
        np.testing.assert_array_equal(seqs, whole.seq_bytes)
    # an empty shard in the middle
    total, ids, _ = stitch_in_process(data, [40, 40], cfg)
    assert total == whole.n_records


def test_record_spanning_a_whole_shard_is_refused():
    data = np.frombuffer(b"@r1\n" + b"A" * 100 + b"\n+\n" + b"!" * 100 + b"\n", dtype=np.uint8)
    shards = [data[:50], data[50:60], data[60:]]
    with pytest.raises(ValueError, match="spans more than one whole shard"):
        sharded.plan_shards([summary_of(s) for s in shards])


def _worker(rank, world, port, data_bytes, cuts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = np.frombuffer(data_bytes, dtype=np.uint8)
    bounds = [0, *cuts, data.size]
    mine = data[bounds[rank]:bounds[rank + 1]]
    n = mine.size
    shard = torch.zeros(n + 4096, dtype=torch.uint8)
    shard[:n] = torch.from_numpy(mine.copy())
    summaries = sharded.gather_summaries(summary_of(mine), torch.device("cpu"))
    plan = sharded.plan_shards(summaries)[rank]
    sharded.exchange_halo(shard, n, plan)
    cfg = O.make_config()
    f = parse_shard(shard[:n + plan.halo_bytes].numpy(), plan, rank == world - 1, cfg)
    before = sharded.records_before(f.n_records, torch.device("cpu"))
    totals, first_err = sharded.reduce_counts(f.n_records, int(f.seq_bytes.size), n, sharded.NO_ERROR, torch.device("cpu"))
    # the one-collective form used by parse_sharded: same totals / offsets; a (pretend) failing local record 5 on the
    # last rank comes out as a global index
    t2, e2, b2 = sharded.gather_outcomes(f.n_records, int(f.seq_bytes.size), n, 5 if rank == world - 1 else -1, torch.device("cpu"))
    assert t2 == totals and b2 == before and (rank != world - 1 or e2 == before + 5)
    q.put((rank, f.n_records, before, totals, (first_err, e2), f.id_bytes.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_exchange():
    rng = np.random.default_rng(5)
    data = rand_stream(rng, n_records=300, max_len=60, dirty=0.0, tail=0)
    whole = O.flat_parse(data, O.make_config())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    cut = len(data) // 2 + 7
    procs = [ctx.Process(target=_worker, args=(r, 2, port, data, [cut], q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, b0, tot0, e0, id0), (r1, n1, b1, tot1, e1, id1) = got
    assert n0 + n1 == whole.n_records and b0 == 0 and b1 == n0
    assert tot0 == tot1 == [whole.n_records, int(whole.seq_bytes.size), len(data)]
    assert e0 == e1 == (sharded.NO_ERROR, n0 + 5)
    assert id0 + id1 == whole.id_bytes.tobytes()
