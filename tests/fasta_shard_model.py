"""The FASTA byte-range shard protocol restated on the host (shared by the CPU planner test and the GPU tests): what
k_fa_probe_headers / k_fa_probe_edges report about a range, and what bzq_fasta_shard_stitch derives from the owners'
outcomes -- with oracle/fasta.py's flat parse standing in for the device parse of a region."""
import ctypes as C

import numpy as np

from blazeseq_amd import _lib as L
from oracle import fasta as FO

SPACES = frozenset([9, 10, 11, 12, 13, 28, 29, 30, 32])
EOF_, TOO_LONG, EMPTY_SEQ, ASCII_BAD = 6, 8, 11, 4


def summary_of(piece: np.ndarray, walk_cap: int = 1 << 62):
    """(n_bytes, first_header, lead_kind, last_byte, tail_open) of one range, byte by byte.  walk_cap: the device stops looking
    for a '>' line's start after line_capacity bytes of spaces (such a line fails wherever it is parsed)."""
    b = piece.tobytes()
    n = len(b)
    first_header = -1
    for q in range(n):
        if b[q] == 62:   # '>'
            j = q - 1
            while j >= 0 and j > q - 1 - walk_cap and b[j] != 10 and b[j] in SPACES:
                j -= 1
            if j >= 0 and b[j] == 10:
                first_header = j + 1
                break
    # the two edge scans look at whole 4 KiB spans and give up once they are line_capacity bytes in
    reach = (walk_cap // 4096 + 1) * 4096 if walk_cap < (1 << 61) else n
    lead_kind = 3
    for q in range(min(n, reach)):
        if b[q] == 10:
            lead_kind = 2
            break
        if b[q] not in SPACES:
            lead_kind = 1 if b[q] == 62 else 0
            break
    tail_open = -1
    lo = max(0, n - reach)
    ln = b.rfind(b"\n", lo)
    if ln >= 0 and ln + 1 < n and all(c in SPACES for c in b[ln + 1:]):
        tail_open = ln + 1
    return (n, first_header, lead_kind, b[-1] if n else 10, tail_open)


def c_plan(summaries):
    P = len(summaries)
    arr = (L.BzqFastaShardSummary * P)()
    for r, s in enumerate(summaries):
        arr[r].n_bytes, arr[r].first_header, arr[r].lead_kind, arr[r].last_byte, arr[r].tail_open = s
    out = (L.BzqFastaShardPlan * P)()
    assert L.lib().bzq_fasta_plan_shards(arr, P, out) == 0
    return list(out)


def count_headers(region: bytes) -> int:
    return sum(1 for ln in region.split(b"\n") if ln.strip(bytes(SPACES)).startswith(b">"))


def oracle_region(check_ascii, line_cap):
    def parse(region, pos_base, record_base=0, line_base=0):
        return FO.flat_parse(region, check_ascii=check_ascii, line_cap=line_cap, is_eof=True, pos_base=pos_base, record_base=record_base,
                             line_base=line_base)
    return parse


def stitch(data: np.ndarray, cuts, check_ascii=False, line_cap=FO.DEFAULT_CAPACITY, summarize=summary_of, parse_region=None):
    """-> (records [(id, seq)], status, message) as bzq_fasta_shard_stitch delivers them over all ranks.  summarize /
    parse_region: the device's probe and parse instead of the restatement and the oracle (tests/fuzz_campaign_fasta_shards.py)."""
    parse_region = parse_region or oracle_region(check_ascii, line_cap)
    bounds = [0, *cuts, data.size]
    shards = [data[bounds[i]:bounds[i + 1]] for i in range(len(bounds) - 1)]
    P = len(shards)
    plans = c_plan([summarize(s) for s in shards])
    regions, flats, owners = [None] * P, [None] * P, []
    for r, (s, p) in enumerate(zip(shards, plans)):
        assert p.stream_pos == bounds[r]
        if s.size == 0 or p.head_bytes >= s.size:
            continue
        halo = [shards[q][:plans[q].head_bytes] for q in range(r + 1, P) if plans[q].head_bytes > 0 and plans[q].head_dst == r]
        assert sum(h.size for h in halo) == p.halo_bytes
        regions[r] = np.concatenate([s[p.head_bytes:], *halo])
        flats[r] = parse_region(regions[r], bounds[r] + p.head_bytes)
        owners.append(r)
    n_rec = [0] * P
    err_rank, prev = -1, -1

    def first_line_too_long(r):
        return prev >= 0 and flats[r].status == TOO_LONG and flats[r].err_record == -1

    for i, r in enumerate(owners):
        f = flats[r]
        n_rec[r] = f.n_records
        if f.status == EOF_:
            prev = r
            continue
        if first_line_too_long(r):
            n_rec[prev] -= 1
        err_rank = r
        if f.status in (EMPTY_SEQ, ASCII_BAD) and f.err_record == count_headers(regions[r].tobytes()) - 1 and i + 1 < len(owners):
            prev = r
            if first_line_too_long(owners[i + 1]):
                err_rank = owners[i + 1]
        break
    records = []
    for r in owners:
        if err_rank >= 0 and r > err_rank:
            break
        records += flats[r].records()[:n_rec[r]]
    if err_rank < 0:
        return records, EOF_, ""
    f = flats[err_rank]
    if f.status != TOO_LONG:
        start = bounds[err_rank] + plans[err_rank].head_bytes
        f = parse_region(regions[err_rank], start, record_base=len(records) - f.n_records,
                         line_base=int(np.count_nonzero(data[:start] == 10)))
    return records, f.status, f.message
