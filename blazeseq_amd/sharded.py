"""Multi-GPU byte-range sharding of one FASTQ stream (new design; the reference is single process,
SURVEY.md 8e).

The file is cut into P contiguous BYTE ranges (not record aligned), one per rank/GPU.  Strict 4-line
framing means the only global dependency is the line index of each shard's first byte, so the
exchange is tiny and latency bound:

  1. every rank scans its shard once (``bzq_shard_scan``): newline count, offsets of its first four
     newlines, first/last byte                                   -> all_gather of 8 x int64 per rank
  2. ownership: a record belongs to the rank that holds its header-line start.  Rank r+1's leading
     bytes up to the end of the straddling record (``head``) are sent to rank r, which appends them
     behind its own bytes (``halo``)                             -> one send/recv per neighbour pair
  3. every rank parses [its bytes + halo] with ``bzq_submit_shard``; counts and the first failing
     record come from one more all_gather of 4 x int64 per rank

No bulk data ever crosses xGMI.  The collectives go through ``torch.distributed`` (backend "nccl" is
RCCL on ROCm; the same code runs over "gloo" on CPU tensors, which is how the protocol is tested
without GPUs).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

SUMMARY_WORDS = 8  # n_bytes, n_newlines, first_nl[4], first_byte, last_byte


@dataclass
class ShardPlan:
    lines_before: int      # global line index of this shard's first (possibly partial) line
    prev_last_byte: int    # byte preceding the shard in the stream (10 for the first shard)
    head_bytes: int        # leading bytes owned by the previous rank's last record
    halo_bytes: int        # bytes to append from the next non-empty shard
    halo_src: int          # rank that sends the halo (-1: none)
    head_dst: int          # rank that receives our head (-1: none)
    records_before: int = 0  # filled after parsing (exclusive scan of per-rank record counts)


def head_bytes_of(summary: Sequence[int], lines_before: int, prev_last_byte: int) -> int:
    """Mirror of bzq_shard_head_bytes (include/blazeseq_hip.h): bytes up to and including the newline
    that ends the record which started in an earlier shard.  -1: the record does not end here."""
    n_bytes, first_nl = summary[0], summary[2:6]
    if n_bytes == 0:
        return 0
    p0 = lines_before & 3
    if prev_last_byte == 10 and p0 == 0:
        return 0
    k = 4 - p0
    if first_nl[k - 1] < 0:
        return -1
    return int(first_nl[k - 1]) + 1


def plan_shards(summaries: Sequence[Sequence[int]]) -> List[ShardPlan]:
    """Pure function of the gathered summaries: what every rank must skip, send and receive."""
    P = len(summaries)
    plans: List[ShardPlan] = []
    lines = 0
    prev_byte = 10
    for r in range(P):
        s = summaries[r]
        hb = head_bytes_of(s, lines, prev_byte)
        if hb < 0:
            raise ValueError(f"shard {r}: a record spans more than one whole shard; use fewer/larger shards")
        plans.append(ShardPlan(lines, prev_byte, hb, 0, -1, -1))
        if s[0] > 0:
            lines += int(s[1])
            prev_byte = int(s[7])
    # the halo of rank r is the head of the next non-empty shard
    for r in range(P):
        if summaries[r][0] == 0:
            continue
        nxt = next((q for q in range(r + 1, P) if summaries[q][0] > 0), -1)
        if nxt >= 0 and plans[nxt].head_bytes > 0:
            plans[r].halo_bytes = plans[nxt].head_bytes
            plans[r].halo_src = nxt
            plans[nxt].head_dst = r
    return plans


def gather_summaries(local: Sequence[int], device, group=None) -> List[List[int]]:
    """all_gather of the 8-word shard summaries: one collective into a [world, 8] tensor, one copy to the host."""
    return _gather_rows(local, device, group)


def exchange_halo(shard: torch.Tensor, n: int, plan: ShardPlan, group=None) -> None:
    """Send our head to the previous owner, receive our halo behind our own bytes.
    ``shard`` is a uint8 tensor with at least n + plan.halo_bytes elements."""
    ops = []
    # gloo moves host tensors: a device-resident shard is staged through the host (bytes to kilobytes)
    staged = shard.is_cuda and dist.get_backend(group) == "gloo"
    recv_buf = None
    if plan.head_dst >= 0 and plan.head_bytes > 0:
        head = shard[:plan.head_bytes]
        ops.append(dist.P2POp(dist.isend, head.cpu() if staged else head, plan.head_dst, group))
    if plan.halo_src >= 0 and plan.halo_bytes > 0:
        if shard.numel() < n + plan.halo_bytes:
            raise ValueError("shard tensor has no room for the halo")
        recv_buf = torch.empty(plan.halo_bytes, dtype=torch.uint8) if staged else shard[n:n + plan.halo_bytes]
        ops.append(dist.P2POp(dist.irecv, recv_buf, plan.halo_src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged and recv_buf is not None:
        shard[n:n + plan.halo_bytes].copy_(recv_buf)
        torch.cuda.synchronize(shard.device)   # the library's own stream reads it next


_gather_bufs = {}


def _gather_rows(row: Sequence[int], device, group=None) -> List[List[int]]:
    """One all_gather_into_tensor of a few int64 per rank and one copy to the host.  The exchange is pure latency, so on
    a GPU the four small tensors (pinned staging on both sides) are made once per (device, width, world) -- 35 us per
    call instead of 52 us with a fresh tensor and a pageable copy back (scripts/gather_ab.py)."""
    world = dist.get_world_size(group)
    device = torch.device(device)
    if device.type != "cuda":   # gloo on CPU tensors (the protocol tests)
        mine = torch.tensor(list(row), dtype=torch.int64)
        out = torch.empty(world * len(row), dtype=torch.int64)
        dist.all_gather_into_tensor(out, mine, group=group)
        return out.view(world, len(row)).tolist()
    key = (device.index, len(row), world, id(group))
    bufs = _gather_bufs.get(key)
    if bufs is None:
        h_in = torch.empty(len(row), dtype=torch.int64, pin_memory=True)
        h_out = torch.empty(world * len(row), dtype=torch.int64, pin_memory=True)
        bufs = (h_in, h_in.numpy(), torch.empty(len(row), dtype=torch.int64, device=device),
                torch.empty(world * len(row), dtype=torch.int64, device=device), h_out, h_out.numpy())
        _gather_bufs[key] = bufs
    h_in, h_in_np, d_in, d_out, h_out, h_out_np = bufs
    h_in_np[:] = list(row)
    d_in.copy_(h_in, non_blocking=True)
    dist.all_gather_into_tensor(d_out, d_in, group=group)
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.current_stream(device).synchronize()
    return h_out_np.reshape(world, len(row)).tolist()


def reduce_counts(records: int, bases: int, nbytes: int, first_error_global: int, device, group=None):
    """Global totals (sum) and the first failing record over all ranks (min of the global indices; 2^62 = none)."""
    rows = _gather_rows([records, bases, nbytes, first_error_global], device, group)
    return [sum(r[k] for r in rows) for k in range(3)], min(r[3] for r in rows)


def gather_outcomes(records: int, bases: int, nbytes: int, first_error_local: int, device, group=None):
    """One all_gather of (records, bases, bytes, first failing LOCAL record or -1) per rank and one copy to the host;
    everything global is derived from it: totals, this rank's records_before (the global index of its record 0) and
    the first failing record over all ranks as a global index (NO_ERROR = none)."""
    rank = dist.get_rank(group)
    rows = _gather_rows([records, bases, nbytes, first_error_local], device, group)
    totals = [sum(r[k] for r in rows) for k in range(3)]
    before, first_err, acc = 0, NO_ERROR, 0
    for q, r in enumerate(rows):
        if q == rank:
            before = acc
        if r[3] >= 0 and first_err == NO_ERROR:
            first_err = acc + r[3]
        acc += r[0]
    return totals, first_err, before


def records_before(n_records: int, device, group=None) -> int:
    return gather_outcomes(n_records, 0, 0, -1, device, group)[2]


NO_ERROR = 1 << 62


def parse_sharded(ctx, shard: torch.Tensor, n: int, stream_pos: int, group=None):
    """Full protocol for one rank on a device-resident shard.  Returns (ChunkResult, plan, totals,
    first_error_global_record)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    coll_device = "cpu" if dist.get_backend(group) == "gloo" else shard.device   # where the small gathers live
    s = ctx.shard_scan(shard.data_ptr(), n)
    local = [int(s.n_bytes), int(s.n_newlines), *[int(x) for x in s.first_nl], int(s.first_byte), int(s.last_byte)]
    summaries = gather_summaries(local, coll_device, group)
    plan = plan_shards(summaries)[rank]
    exchange_halo(shard, n, plan, group)
    is_last = all(summaries[q][0] == 0 for q in range(rank + 1, world))
    ctx.submit_shard(shard.data_ptr(), n, plan.halo_bytes, plan.lines_before, plan.prev_last_byte, stream_pos, is_last)
    res = ctx.result()
    err_local = int(res.error_record) if res.status not in (0, 6) else -1
    totals, first_err, before = gather_outcomes(int(res.n_records), int(res.seq_bytes), n, err_local, coll_device, group)
    plan.records_before = before
    return res, plan, totals, first_err
