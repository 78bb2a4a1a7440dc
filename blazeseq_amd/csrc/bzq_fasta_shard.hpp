// bzq_fasta_shard.hpp -- the FASTA parser over byte-range shards: bzq_fasta_plan_shards / bzq_fasta_shard_stitch
// (include/blazeseq_hip.h "FASTA over byte-range shards").  New design; the reference is one sequential FastaParser
// (blazeseq/fasta/parser.mojo:122-203 over LineIterator, blazeseq/io/buffered.mojo:600-638).
//
// Ownership: a record belongs to the rank in whose byte range its header LINE starts.  The kernels of bzq_fasta.hip parse
// [a rank's first header line, the next owner's first header line) as one complete stream, so nothing about them changes;
// what is new is finding those cuts without a sequential pass (k_fa_probe_headers / k_fa_probe_edges, bzq_fasta.hpp) and
// the two places where the sequential parser's order of events crosses a cut:
//   * a header line of >= line_capacity bytes raises while the record BEFORE it is still open (next_line comes before the
//     record is closed, parser.mojo:135-170): when an owner's first line is too long, the owner before it loses its last
//     record, and a validation error of that last record is never reached;
//   * error text carries stream-global record / line numbers: on the error path only, the ranks count their newlines and
//     the failing rank parses again with the right bases.
// Transports are those of the FASTQ protocol (bzq_comm.hpp): RCCL, or POSIX shared memory for same-node ranks.
//
// Included by bzq_api.hip inside its extern "C" block, behind bzq_comm.hpp.
#pragma once

int32_t bzq_fasta_shard_probe_(bzq_fasta* h, const uint8_t* d, uint64_t n, int64_t row[5]);
int32_t bzq_fasta_count_newlines_(bzq_fasta* h, const uint8_t* d, uint64_t n, int64_t* out);
int64_t bzq_fasta_last_headers_(const bzq_fasta* h);

int32_t bzq_fasta_plan_shards(const bzq_fasta_shard_summary* all, int32_t nranks, bzq_fasta_shard_plan* out) {
    if (!all || !out || nranks <= 0) return BZQ_ERR_ARG;
    std::vector<int64_t> cut((size_t)nranks, -1);   // start of the first header line that begins in the rank's range
    int pend_rank = -1;      // a line start whose first non-space byte has not been seen yet: (rank, offset)
    int64_t pend_pos = 0;
    bool at_next = true;     // the next non-empty range begins a line (stream start, or the byte before it is '\n')
    int first = -1;
    uint64_t pos = 0;
    for (int r = 0; r < nranks; ++r) {
        const bzq_fasta_shard_summary& s = all[r];
        bzq_fasta_shard_plan& p = out[r];
        memset(&p, 0, sizeof(p));
        p.stream_pos = pos; p.head_dst = -1; p.halo_first_src = -1;
        pos += s.n_bytes;
        if (s.n_bytes == 0) continue;
        if (first < 0) first = r;
        if (at_next) { pend_rank = r; pend_pos = 0; at_next = false; }
        if (s.lead_kind == 3) continue;   // only spaces and no '\n': whatever was open stays open
        if (pend_rank >= 0) {
            if (s.lead_kind == 1 && (cut[(size_t)pend_rank] < 0 || pend_pos < cut[(size_t)pend_rank])) cut[(size_t)pend_rank] = pend_pos;
            pend_rank = -1;
        }
        if (s.first_header >= 0 && (cut[(size_t)r] < 0 || s.first_header < cut[(size_t)r])) cut[(size_t)r] = s.first_header;
        if (s.tail_open >= 0) { pend_rank = r; pend_pos = s.tail_open; }
        else if (s.last_byte == 10) at_next = true;
    }
    if (first >= 0) cut[(size_t)first] = 0;   // the stream's first rank parses from offset 0 whatever is there
    int owner = -1, last_owner = -1;
    for (int r = 0; r < nranks; ++r) {
        const bzq_fasta_shard_summary& s = all[r];
        bzq_fasta_shard_plan& p = out[r];
        if (s.n_bytes == 0) continue;
        const uint64_t head = cut[(size_t)r] >= 0 ? (uint64_t)cut[(size_t)r] : s.n_bytes;
        p.head_bytes = head;
        if (head > 0) {
            if (owner < 0) return BZQ_ERR_ARG;   // cannot happen: the first non-empty rank has no head
            bzq_fasta_shard_plan& o = out[owner];
            p.head_dst = owner;
            p.halo_offset = o.halo_bytes;
            o.halo_bytes += head;
            if (o.halo_first_src < 0) o.halo_first_src = r;
            o.halo_n_src = r - o.halo_first_src + 1;
        }
        if (head < s.n_bytes) { owner = r; last_owner = r; }
    }
    out[last_owner >= 0 ? last_owner : 0].is_last = 1;
    return 0;
}

// Failures are collective, as in bzq_shard_stitch (bzq_comm.hpp): between the first and the last exchange a rank never returns
// on its own -- its failure rides in the next gathered row (word 7) and every rank returns together behind that gather.
int32_t bzq_fasta_shard_stitch(bzq_ctx* c, bzq_fasta* h, uint8_t* d_shard, uint64_t n, uint64_t capacity, bzq_fasta_shard_result* out) {
    if (!h || !out) return BZQ_ERR_ARG;
    bzq_comm* m = c ? c->comm : nullptr;
    const int P = m ? m->nranks : 1, me = m ? m->rank : 0;
    auto fail = [&](const std::string& msg, int32_t code) { bzq_fasta_set_error_(h, msg.c_str()); return code; };
    auto gather = [&](const int64_t* row, int64_t* all) -> int {
        if (!m) { memcpy(all, row, COMM_ROW * 8); return 0; }
        const int rc = comm_gather(c, row, all);
        if (rc) bzq_fasta_set_error_(h, c->err.c_str());
        return rc;
    };
    memset(out, 0, sizeof(*out));
    out->first_error_record = -1; out->error_rank = -1;
    int rc;
    int lrc = 0;          // this rank's own failure so far
    std::string lerr;     // ... and its text
    auto note = [&](int r) { if (r < 0 && !lrc) { lrc = r; lerr = bzq_fasta_last_error(h); } return r; };
    auto note_msg = [&](int code, const std::string& msg) { if (!lrc) { lrc = code; lerr = msg; } };
    auto note_hip = [&](hipError_t e, const char* what) { if (e != hipSuccess) note_msg(BZQ_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
    auto everybody_fails = [&](const std::vector<int64_t>& rows, const char* where) -> int {
        for (int r = 0; r < P; ++r) {
            const int64_t code = rows[(size_t)r * COMM_ROW + 7];
            if (code >= 0) continue;
            if (r == me && lrc) return fail(lerr, lrc);
            return fail("bzq_fasta_shard_stitch: rank " + std::to_string(r) + " failed (" + std::to_string(code) + ") " + where, BZQ_ERR_IO);
        }
        return 0;
    };
    if ((!d_shard && n) || capacity < n) note_msg(BZQ_ERR_ARG, "bzq_fasta_shard_stitch: bad shard buffer");
    else if (m && c->device != bzq_fasta_device_(h)) note_msg(BZQ_ERR_ARG, "bzq_fasta_shard_stitch: the communicator's ctx and the FASTA handle are on different devices");

    // 1. probe + summary all-gather: {bytes, first header, lead kind, tail open, last byte, room, -, failure}
    int64_t row[COMM_ROW] = {0, -1, 3, -1, 10, 0, 0, 0};
    if (!lrc && note(bzq_fasta_shard_probe_(h, d_shard, n, row)) < 0) { const int64_t empty[COMM_ROW] = {0, -1, 3, -1, 10, 0, 0, 0}; memcpy(row, empty, sizeof(row)); }
    if (!lrc) row[5] = (int64_t)(capacity - n);   // room behind the range's bytes: every rank checks every rank's halo against it
    row[6] = 0; row[7] = lrc;
    std::vector<int64_t> all((size_t)P * COMM_ROW);
    if ((rc = gather(row, all.data()))) return rc;
    if ((rc = everybody_fails(all, "before the ranges were exchanged"))) return rc;
    std::vector<bzq_fasta_shard_summary> sums((size_t)P);
    for (int r = 0; r < P; ++r) {
        const int64_t* w = &all[(size_t)r * COMM_ROW];
        sums[(size_t)r] = bzq_fasta_shard_summary{(uint64_t)w[0], w[1], (int32_t)w[2], (int32_t)w[4], w[3]};
    }

    // 2. plan: a pure function of the gathered rows -- whatever it refuses, it refuses on every rank
    std::vector<bzq_fasta_shard_plan> plans((size_t)P);
    if ((rc = bzq_fasta_plan_shards(sums.data(), P, plans.data()))) return fail("bzq_fasta_shard_stitch: inconsistent shard summaries", rc);
    const bzq_fasta_shard_plan pl = plans[(size_t)me];
    out->plan = pl;
    for (int r = 0; r < P; ++r)   // fails on ALL ranks, before anything is exchanged
        if ((int64_t)plans[(size_t)r].halo_bytes > all[(size_t)r * COMM_ROW + 5])
            return fail("bzq_fasta_shard_stitch: rank " + std::to_string(r) + "'s shard buffer has no room for its halo (" + std::to_string(plans[(size_t)r].halo_bytes) +
                        " bytes behind " + std::to_string(sums[(size_t)r].n_bytes) + ", room for " + std::to_string(all[(size_t)r * COMM_ROW + 5]) + ")", BZQ_ERR_ARG);
    if (m && P > 1 && m->kind == 2)
        for (int r = 0; r < P; ++r)   // (the capacity is the same on every rank: all of them fail together)
            if (plans[(size_t)r].head_bytes > m->halo_cap)
                return fail("bzq_fasta_shard_stitch(shm): rank " + std::to_string(r) + "'s head of " + std::to_string(plans[(size_t)r].head_bytes) + " bytes exceeds the halo capacity the communicator was created with", BZQ_ERR_ARG);

    // 3. heads travel to their owners (comm_exchange_heads, bzq_comm.hpp: the FASTQ protocol's exchange, deadline included)
    if (m && P > 1) {
        note_hip(hipSetDevice(c->device), "hipSetDevice");
        int xl = 0;
        std::string xe;
        if ((rc = comm_exchange_heads(c, m, d_shard, n, plans, xl, xe))) return fail(c->err, rc);
        if (xl) note_msg(xl, xe);
    }

    // 4. every owner parses [its first header line, end of range + halo) as one complete stream
    const bool owner = n > 0 && pl.head_bytes < n;
    const uint8_t* region = d_shard + pl.head_bytes;
    const uint64_t region_n = owner ? n - pl.head_bytes + pl.halo_bytes : 0;
    bzq_fasta_chunk res{};
    res.status = BZQ_EOF;
    if (owner && !lrc && note(bzq_fasta_parse(h, region, region_n, 1, pl.stream_pos + pl.head_bytes, 0, 0, &res)) < 0) { res = bzq_fasta_chunk{}; res.status = BZQ_EOF; }

    // 5. outcomes: {records, status, headers, open record at the error (-2 none), owner, -, -, runtime failure}
    const bool failed = owner && res.status != BZQ_EOF;
    int64_t orow[COMM_ROW] = {(int64_t)res.n_records, res.status, owner && !lrc ? bzq_fasta_last_headers_(h) : 0, failed ? bzq_fasta_error_open_record(h) : -2, owner ? 1 : 0, 0, 0, lrc};
    std::vector<int64_t> oc((size_t)P * COMM_ROW);
    if ((rc = gather(orow, oc.data()))) return rc;
    if ((rc = everybody_fails(oc, "while parsing its range"))) return rc;
    std::vector<int64_t> n_rec((size_t)P, 0);
    int err_rank = -1, prev_owner = -1;
    auto first_line_too_long = [&](int r) {   // rank r's first line (a header line) is too long: the record before it was still open
        const int64_t* w = &oc[(size_t)r * COMM_ROW];
        return prev_owner >= 0 && w[1] == BZQ_BUFFER_EXCEEDED && w[3] == -1;
    };
    for (int r = 0; r < P && err_rank < 0; ++r) {
        const int64_t* w = &oc[(size_t)r * COMM_ROW];
        if (!w[4]) continue;
        n_rec[(size_t)r] = w[0];
        if (w[1] == BZQ_EOF) { prev_owner = r; continue; }
        if (first_line_too_long(r)) n_rec[(size_t)prev_owner] -= 1;   // prev_owner parsed clean: its last record is the one that was open
        err_rank = r;
        if ((w[1] == BZQ_FASTA_EMPTY_SEQUENCE || w[1] == BZQ_ASCII_INVALID) && w[3] == w[2] - 1) {
            // the failing record is the rank's last: it is validated when the NEXT header line has been read, and reading
            // that line fails first when it is too long
            int nx = r + 1;
            while (nx < P && !oc[(size_t)nx * COMM_ROW + 4]) ++nx;
            prev_owner = r;
            if (nx < P && first_line_too_long(nx)) err_rank = nx;
        }
    }
    uint64_t before = 0;
    for (int r = 0; r < P; ++r) {
        if (err_rank >= 0 && r > err_rank) n_rec[(size_t)r] = 0;
        if (r == me) out->records_before = before;
        before += (uint64_t)n_rec[(size_t)r];
    }
    out->global_records = before;
    out->stream_status = err_rank >= 0 ? (int32_t)oc[(size_t)err_rank * COMM_ROW + 1] : BZQ_EOF;
    if (err_rank >= 0) { out->first_error_record = (int64_t)before; out->error_rank = err_rank; }

    // 6. cold: stream-global numbers in the error text
    if (err_rank >= 0 && out->stream_status != BZQ_BUFFER_EXCEEDED) {
        int64_t nl[2] = {0, 0};
        if (note(bzq_fasta_count_newlines_(h, d_shard, n, &nl[0])) >= 0) note(bzq_fasta_count_newlines_(h, d_shard, pl.head_bytes, &nl[1]));
        int64_t lrow[COMM_ROW] = {nl[0], nl[1], 0, 0, 0, 0, 0, lrc};
        std::vector<int64_t> la((size_t)P * COMM_ROW);
        if ((rc = gather(lrow, la.data()))) return rc;
        if ((rc = everybody_fails(la, "while counting its lines for the error text"))) return rc;
        if (me == err_rank) {   // (the last exchange is behind us: a failure here is this rank's alone)
            uint64_t line_base = (uint64_t)nl[1];
            for (int r = 0; r < me; ++r) line_base += (uint64_t)la[(size_t)r * COMM_ROW];
            if ((rc = bzq_fasta_parse(h, region, region_n, 1, pl.stream_pos + pl.head_bytes, line_base, out->records_before, &res))) return rc;
        }
    }

    if ((int64_t)res.n_records != n_rec[(size_t)me]) {   // the last record was still open when a later rank's header line failed, or the stream ended before this rank
        res.n_records = n_rec[(size_t)me];
        res.seq_bytes = res.id_bytes = 0;
        if (res.n_records > 0) {
            int64_t e[2];
            if ((rc = bzq_fasta_copy_to_host(h, &e[0], res.d_seq_ends + (res.n_records - 1), 8)) ||
                (rc = bzq_fasta_copy_to_host(h, &e[1], res.d_id_ends + (res.n_records - 1), 8)))
                return rc;
            res.seq_bytes = e[0]; res.id_bytes = e[1];
        }
    }
    out->chunk = res;
    return 0;
}
