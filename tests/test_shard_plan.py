"""bzq_plan_shards (C ABI, blazeseq_amd/csrc/bzq_comm.hpp) without a GPU: the planner is a pure function of the gathered
shard summaries.  Checked against the Python planner of blazeseq_amd/sharded.py (the cross-check implementation) and,
for records longer than a whole shard -- which sharded.py refuses -- against a brute-force walk over the bytes."""
import ctypes as C

import numpy as np
import pytest

from blazeseq_amd import _lib as L
from blazeseq_amd import sharded
from tests.fastq_fuzz import rand_stream


def summarize(piece: np.ndarray):
    nl = np.flatnonzero(piece == 10)
    first = [int(nl[i]) if i < nl.size else -1 for i in range(4)]
    return [int(piece.size), int(nl.size), *first, int(piece[0]) if piece.size else 10, int(piece[-1]) if piece.size else 10]


def c_plan(summaries):
    P = len(summaries)
    arr = (L.BzqShardSummary * P)()
    for r, s in enumerate(summaries):
        arr[r].n_bytes, arr[r].n_newlines = s[0], s[1]
        for i in range(4):
            arr[r].first_nl[i] = s[2 + i]
        arr[r].first_byte, arr[r].last_byte = s[6], s[7]
    out = (L.BzqShardPlan * P)()
    assert L.lib().bzq_plan_shards(arr, P, out) == 0
    return list(out)


def brute_force(data: np.ndarray, cuts):
    """For every shard: line index at its first byte, head bytes, halo bytes, owner of the head -- from the bytes."""
    bounds = [0, *cuts, data.size]
    nl = np.flatnonzero(data == 10)
    rec_end = nl[3::4]                        # offset of every record's terminating newline
    rec_start = np.concatenate([[0], rec_end + 1])  # start of record k (the last entry: start of the unterminated tail)
    plans = []
    for r in range(len(bounds) - 1):
        a, b = bounds[r], bounds[r + 1]
        lines_before = int(np.searchsorted(nl, a))
        # head: bytes of [a, b) before the first record start >= a
        k = int(np.searchsorted(rec_start, a))
        first_start = int(rec_start[k]) if k < rec_start.size else data.size
        if k == rec_start.size - 1 and rec_end.size == rec_start.size - 1 and first_start >= data.size:
            first_start = data.size
        head = min(b, first_start) - a if b > a else 0
        plans.append(dict(lines_before=lines_before, head=head, n=b - a))
    # owners: a rank owns records iff head < n; halo = heads of the following ranks up to the next owner
    for r, p in enumerate(plans):
        p["halo"] = 0
        if p["n"] and p["head"] < p["n"]:
            q = r + 1
            while q < len(plans) and not (plans[q]["n"] and plans[q]["head"] < plans[q]["n"]):
                p["halo"] += plans[q]["head"]
                q += 1
            if q < len(plans):
                p["halo"] += plans[q]["head"]
    return plans


@pytest.mark.parametrize("seed", range(40))
def test_c_planner_equals_python_planner_and_the_bytes(seed):
    rng = np.random.default_rng(seed)
    data = np.frombuffer(rand_stream(rng, n_records=int(rng.integers(3, 60)), max_len=int(rng.choice([8, 40, 300])), dirty=0.0,
                                     tail=int(rng.choice([0, 1, 5]))), dtype=np.uint8)
    P = int(rng.integers(1, 9))
    cuts = sorted(int(x) for x in rng.integers(0, data.size + 1, size=P - 1))
    bounds = [0, *cuts, data.size]
    sums = [summarize(data[bounds[r]:bounds[r + 1]]) for r in range(P)]
    cp = c_plan(sums)
    bf = brute_force(data, cuts)
    for r in range(P):
        assert cp[r].lines_before == bf[r]["lines_before"] or sums[r][0] == 0, (seed, r)
        assert cp[r].head_bytes == bf[r]["head"], (seed, r, cp[r].head_bytes, bf[r])
        assert cp[r].halo_bytes == bf[r]["halo"], (seed, r)
    try:
        pp = sharded.plan_shards(sums)
    except ValueError:
        assert any(cp[r].head_bytes == sums[r][0] and sums[r][0] > 0 for r in range(P))   # a record longer than a shard
        return
    for r in range(P):
        assert (cp[r].lines_before, cp[r].prev_last_byte, cp[r].head_bytes, cp[r].halo_bytes) == \
               (pp[r].lines_before, pp[r].prev_last_byte, pp[r].head_bytes, pp[r].halo_bytes), (seed, r)
        assert cp[r].head_dst == pp[r].head_dst
        if pp[r].halo_src >= 0:
            assert cp[r].halo_first_src <= pp[r].halo_src < cp[r].halo_first_src + cp[r].halo_n_src


def test_record_longer_than_whole_shards():
    rec = b"@r1\n" + b"A" * 100 + b"\n+\n" + b"I" * 100 + b"\n"
    data = np.frombuffer(b"@r0\nAC\n+\nII\n" + rec + b"@r2\nG\n+\nI\n", dtype=np.uint8)
    cuts = [20, 50, 80, 150, 215]       # shards 1..3 lie inside r1
    bounds = [0, *cuts, data.size]
    sums = [summarize(data[bounds[r]:bounds[r + 1]]) for r in range(len(bounds) - 1)]
    cp = c_plan(sums)
    bf = brute_force(data, cuts)
    assert [p.head_bytes for p in cp] == [b["head"] for b in bf]
    assert [p.halo_bytes for p in cp] == [b["halo"] for b in bf]
    owners = [r for r, p in enumerate(cp) if sums[r][0] and p.head_bytes < sums[r][0]]
    assert cp[owners[-1]].is_last == 1 and sum(p.is_last for p in cp) == 1
    assert cp[2].head_bytes == sums[2][0] and cp[2].head_dst == cp[1].head_dst or cp[2].head_dst == 1
    # the heads of one owner tile its halo without gaps, in rank order
    for o in owners:
        off = 0
        for q in range(len(cp)):
            if cp[q].head_dst == o and cp[q].head_bytes:
                assert cp[q].halo_offset == off
                off += cp[q].head_bytes
        assert off == cp[o].halo_bytes


def test_empty_stream_and_empty_shards():
    cp = c_plan([[0, 0, -1, -1, -1, -1, 10, 10]] * 3)
    assert sum(p.is_last for p in cp) == 1 and all(p.head_bytes == 0 and p.halo_bytes == 0 for p in cp)
