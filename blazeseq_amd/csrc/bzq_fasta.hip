// bzq_fasta.hip -- host side of the FASTA entry points of libblazeseq_hip.so (include/blazeseq_hip.h, "FASTA records").
//
// The handle is the reference's FastaParser for one stream (blazeseq/fasta/parser.mojo:60-120), driven a chunk at a
// time: bzq_fasta_parse is next_record for every record of the chunk at once.  There is NO CPU fallback: without a
// gfx950 device bzq_fasta_create fails.
#include "../../include/blazeseq_hip.h"
#include "bzq_fasta.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

using namespace bzq;
using namespace bzq::fa;

namespace {
thread_local std::string g_fa_create_error;
struct Buf { void* p = nullptr; size_t cap = 0; };
}

struct bzq_fasta {
    int device = 0;
    bzq_fasta_config cfg{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    Buf in, seq, id, seq_ends, id_ends, hdr_pos, sums, tile_in, tile_cnt, grp, base, gen_prefix, probe;
    FaState* d_state = nullptr;
    FaState* h_state = nullptr;   // pinned
    // last chunk
    const uint8_t* cur = nullptr;
    uint64_t cur_n = 0, stream_pos = 0, line_base = 0, record_base = 0;
    int64_t rec_cap = 0, n_tiles = 0;
    bzq_fasta_chunk res{};
    int64_t killed = INT64_MAX;   // last chunk with an error: the record that was open when it was met (-1 = before any header)
    int64_t n_headers = 0;
    std::string message;
};

namespace {

#define FACHK(h, call)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return BZQ_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

int ensure(bzq_fasta* h, Buf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) { FACHK(h, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    const hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        h->err = "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e);
        b.p = nullptr;
        return BZQ_ERR_NOMEM;
    }
    b.cap = want;
    return 0;
}

bool is_device_pointer(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice;
}

// cold path: k_fa_query for one position -> {newlines before, start of its line, headers before}
int query(bzq_fasta* h, int64_t pos, int64_t out[3]) {
    QueryArgs q{h->cur, (int64_t)h->cur_n, (const int64_t*)h->base.p, (const int64_t*)h->hdr_pos.p,
                std::min<int64_t>(h->h_state->n_headers, h->rec_cap), pos, h->d_state, h->cfg.line_capacity};
    hipLaunchKernelGGL(k_fa_query, dim3(1), dim3(BLOCK), 0, h->stream, q);
    FaState tmp;
    FACHK(h, hipMemcpyAsync(&tmp, h->d_state, sizeof tmp, hipMemcpyDeviceToHost, h->stream));
    FACHK(h, hipStreamSynchronize(h->stream));
    out[0] = tmp.query[0]; out[1] = tmp.query[1]; out[2] = tmp.query[2];
    return 0;
}

std::string parse_error_text(const char* msg, int64_t rec, int64_t line, int64_t pos) {   // errors.mojo:178-192
    std::string s = msg;
    if (rec > 0) s += "\n  Record number: " + std::to_string(rec);
    if (line > 0) s += "\n  Line number: " + std::to_string(line);
    if (pos > 0) s += "\n  File position: " + std::to_string(pos);
    return s;
}

} // namespace

extern "C" {

int32_t bzq_fasta_create(int32_t device, const bzq_fasta_config* cfg, bzq_fasta** out) {
    if (!out) return BZQ_ERR_ARG;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
        (void)hipGetLastError();
        g_fa_create_error = "bzq_fasta_create: no HIP device " + std::to_string(device) + " (this library has no CPU fallback)";
        return BZQ_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_fa_create_error = std::string("bzq_fasta_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return BZQ_ERR_NO_DEVICE;
    }
    bzq_fasta* h = new bzq_fasta();
    h->device = device;
    if (cfg) h->cfg = *cfg;
    if (h->cfg.line_capacity == 0) h->cfg.line_capacity = 256 * 1024;
    if (h->cfg.line_capacity < 2 * TILE) {
        g_fa_create_error = "bzq_fasta_create: line_capacity must be 0 (256 KiB) or at least 32768";
        delete h;
        return BZQ_ERR_ARG;
    }
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&h->ev0) == hipSuccess && hipEventCreate(&h->ev1) == hipSuccess &&
              hipMalloc((void**)&h->d_state, sizeof(FaState)) == hipSuccess &&
              hipHostMalloc((void**)&h->h_state, sizeof(FaState), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        g_fa_create_error = "bzq_fasta_create: stream / event / state allocation failed";
        bzq_fasta_destroy(h);
        return BZQ_ERR_HIP;
    }
    *out = h;
    return 0;
}

void bzq_fasta_destroy(bzq_fasta* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (Buf* b : {&h->in, &h->seq, &h->id, &h->seq_ends, &h->id_ends, &h->hdr_pos, &h->sums, &h->tile_in, &h->tile_cnt, &h->grp,
                   &h->base, &h->gen_prefix, &h->probe})
        if (b->p) (void)hipFree(b->p);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* bzq_fasta_last_error(const bzq_fasta* h) { return h ? h->err.c_str() : g_fa_create_error.c_str(); }

// internal (bzq_api.hip: the ingest pipeline in front of this parser)
int32_t bzq_fasta_device_(const bzq_fasta* h) { return h ? h->device : 0; }
void bzq_fasta_set_error_(bzq_fasta* h, const char* msg) { if (h) h->err = msg ? msg : ""; }

int32_t bzq_fasta_copy_to_host(bzq_fasta* h, void* dst, const void* d_src, size_t bytes) {
    if (!h || (bytes && (!dst || !d_src))) return BZQ_ERR_ARG;
    if (!bytes) return 0;
    FACHK(h, hipSetDevice(h->device));
    FACHK(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    FACHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int32_t bzq_fasta_parse(bzq_fasta* h, const uint8_t* data, uint64_t n, int32_t is_eof, uint64_t stream_pos, uint64_t line_base,
                        uint64_t record_base, bzq_fasta_chunk* out) {
    if (!h || !out || (n && !data)) return BZQ_ERR_ARG;
    FACHK(h, hipSetDevice(h->device));
    int rc;
    const int64_t nt = (int64_t)((n + TILE - 1) / TILE);
    const int64_t ng = (nt + BLOCK - 1) / BLOCK;
    int64_t rec_cap = 0;   // sized from the header count once pass 1 and the scan are through (below)
    const bool on_device = n && is_device_pointer(data);
    if ((!on_device && (rc = ensure(h, h->in, (size_t)n + 64))) || (rc = ensure(h, h->seq, (size_t)n + 64)) ||
        (rc = ensure(h, h->id, (size_t)n + 64)) ||
        (rc = ensure(h, h->sums, (size_t)(nt + 1) * 24)) || (rc = ensure(h, h->tile_in, (size_t)(nt + 1) * 4)) ||
        (rc = ensure(h, h->tile_cnt, (size_t)(nt + 1) * 16)) || (rc = ensure(h, h->grp, (size_t)(ng + 1) * 32)) ||
        (rc = ensure(h, h->base, (size_t)(nt + 1) * 32)))
        return rc;
    const uint8_t* d = data;
    if (!on_device && n) {
        FACHK(h, hipMemcpyAsync(h->in.p, data, n, hipMemcpyHostToDevice, h->stream));
        d = (const uint8_t*)h->in.p;
    }
    h->cur = d; h->cur_n = n; h->stream_pos = stream_pos; h->line_base = line_base; h->record_base = record_base;
    h->rec_cap = rec_cap; h->n_tiles = nt;
    h->message.clear();

    FaState init{};
    init.long_pos = init.long_tile = init.nohdr_pos = init.ascii_rec = init.empty_rec = NONE;
    *h->h_state = init;
    FACHK(h, hipMemcpyAsync(h->d_state, h->h_state, sizeof(FaState), hipMemcpyHostToDevice, h->stream));
    FACHK(h, hipEventRecord(h->ev0, h->stream));
    if (nt > 0) {
        SumsArgs sa{d, (int64_t)n, (u64*)h->sums.p};
        hipLaunchKernelGGL(k_fa_tile_sums, dim3((unsigned)nt), dim3(BLOCK), 0, h->stream, sa);
        ResolveArgs ra{(const u64*)h->sums.p, nt, (int64_t)n, is_eof, h->cfg.line_capacity, (uint32_t*)h->tile_in.p,
                       (uint32_t*)h->tile_cnt.p, (int64_t*)h->grp.p, h->d_state};
        hipLaunchKernelGGL(k_fa_resolve, dim3((unsigned)ng), dim3(BLOCK), 0, h->stream, ra);
        BasesArgs ba{(const uint32_t*)h->tile_cnt.p, (const int64_t*)h->grp.p, nt, ng, (int64_t*)h->base.p, h->d_state};
        hipLaunchKernelGGL(k_fa_bases, dim3((unsigned)ng), dim3(BLOCK), 0, h->stream, ba);
        // per-record arrays: a guess first (one record per 64 input bytes, or what the last chunk needed); the kernels
        // never write past rec_cap, and in the rare case the chunk holds more headers pass 2 is repeated with arrays of
        // the right size (a bound from the chunk size alone would be n/4 records = 6 bytes of arrays per input byte)
        rec_cap = std::max<int64_t>({(int64_t)(h->seq_ends.cap / 8), (int64_t)(n / 64) + 1024});
        if ((rc = ensure(h, h->seq_ends, (size_t)rec_cap * 8)) || (rc = ensure(h, h->id_ends, (size_t)rec_cap * 8)) ||
            (rc = ensure(h, h->hdr_pos, (size_t)rec_cap * 8)))
            return rc;
        h->rec_cap = rec_cap;
        auto pass2 = [&]() {
            fa::EmitArgs ea{d, (int64_t)n, (const uint32_t*)h->tile_in.p, (const int64_t*)h->base.p, (uint8_t*)h->seq.p, (uint8_t*)h->id.p,
                            (int64_t*)h->seq_ends.p, (int64_t*)h->id_ends.p, (int64_t*)h->hdr_pos.p, rec_cap, h->d_state};
            if (h->cfg.check_ascii) hipLaunchKernelGGL(k_fa_emit<true>, dim3((unsigned)nt), dim3(BLOCK), 0, h->stream, ea);
            else hipLaunchKernelGGL(k_fa_emit<false>, dim3((unsigned)nt), dim3(BLOCK), 0, h->stream, ea);
            FinishArgs fa{d, (int64_t)n, is_eof, (const u64*)h->sums.p, (const int64_t*)h->base.p, nt, (int64_t*)h->seq_ends.p,
                          (int64_t*)h->id_ends.p, (const int64_t*)h->hdr_pos.p, rec_cap, h->d_state, h->cfg.line_capacity};
            hipLaunchKernelGGL(k_fa_finish, dim3(1), dim3(BLOCK), 0, h->stream, fa);
            EmptyArgs ema{(const int64_t*)h->seq_ends.p, h->d_state, rec_cap};
            hipLaunchKernelGGL(k_fa_empty, dim3(512), dim3(BLOCK), 0, h->stream, ema);
            (void)hipEventRecord(h->ev1, h->stream);
        };
        pass2();
        FACHK(h, hipMemcpyAsync(h->h_state, h->d_state, sizeof(FaState), hipMemcpyDeviceToHost, h->stream));
        FACHK(h, hipStreamSynchronize(h->stream));
        if (h->h_state->n_headers + 2 > rec_cap) {   // more records than guessed: arrays of the right size, pass 2 again
            rec_cap = h->h_state->n_headers + 2;
            if ((rc = ensure(h, h->seq_ends, (size_t)rec_cap * 8)) || (rc = ensure(h, h->id_ends, (size_t)rec_cap * 8)) ||
                (rc = ensure(h, h->hdr_pos, (size_t)rec_cap * 8)))
                return rc;
            h->rec_cap = rec_cap;
            pass2();
        }
    } else {
        FACHK(h, hipEventRecord(h->ev1, h->stream));
    }
    FACHK(h, hipMemcpyAsync(h->h_state, h->d_state, sizeof(FaState), hipMemcpyDeviceToHost, h->stream));
    FACHK(h, hipStreamSynchronize(h->stream));
    FACHK(h, hipGetLastError());
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);

    const FaState& st = *h->h_state;
    bzq_fasta_chunk r{};
    r.kernel_ms = ms;
    r.d_seq_bytes = (const uint8_t*)h->seq.p; r.d_id_bytes = (const uint8_t*)h->id.p;
    r.d_seq_ends = (const int64_t*)h->seq_ends.p; r.d_id_ends = (const int64_t*)h->id_ends.p; r.d_hdr_pos = (const int64_t*)h->hdr_pos.p;
    const int64_t H = st.n_headers, closed = nt ? st.n_closed : 0;
    const int64_t cap = h->cfg.line_capacity;

    // ---- the first error, in the order the reference meets them (oracle/fasta_oracle.c) ----
    // a line of >= capacity bytes: one with its '\n' (k_fa_resolve), or the last line of the input
    int64_t long_pos = st.long_pos == NONE ? -1 : (int64_t)st.long_pos;
    if (st.long_tile != NONE) {   // a line so long that pass 1's bounded walk lost its start: find it (cold)
        LongStartArgs la{(const u64*)h->sums.p, (int64_t)st.long_tile, h->d_state};
        hipLaunchKernelGGL(k_fa_long_start, dim3(1), dim3(BLOCK), 0, h->stream, la);
        FaState tmp;
        FACHK(h, hipMemcpyAsync(&tmp, h->d_state, sizeof tmp, hipMemcpyDeviceToHost, h->stream));
        FACHK(h, hipStreamSynchronize(h->stream));
        if (long_pos < 0 || tmp.query[3] < long_pos) long_pos = tmp.query[3];
    }
    if (nt && is_eof && (int64_t)n - st.last_line_start >= cap && (long_pos < 0 || st.last_line_start < long_pos)) long_pos = st.last_line_start;
    int64_t killed = INT64_MAX;   // the record that is open when the error is met; -1 = before any header
    int code = 0;
    if (long_pos >= 0) {
        int64_t q[3];
        if ((rc = query(h, long_pos, q))) return rc;
        killed = q[2] - 1; code = BZQ_BUFFER_EXCEEDED;
    }
    // (in a chunk that is not the last, a line without its '\n' yet is not judged: it may still turn out too long)
    const bool nohdr = st.nohdr_pos != NONE && (is_eof || (int64_t)st.nohdr_pos < st.last_line_start);
    if (nohdr && !(code == BZQ_BUFFER_EXCEEDED && long_pos <= (int64_t)st.nohdr_pos)) { killed = -1; code = BZQ_FASTA_NO_HEADER; }
    {
        int64_t rec = INT64_MAX; int rcode = 0;
        if (st.empty_rec != NONE && (int64_t)st.empty_rec < closed) { rec = (int64_t)st.empty_rec; rcode = BZQ_FASTA_EMPTY_SEQUENCE; }
        if (st.ascii_rec != NONE && (int64_t)st.ascii_rec < closed && (int64_t)st.ascii_rec < rec) { rec = (int64_t)st.ascii_rec; rcode = BZQ_ASCII_INVALID; }
        if (rcode && rec < killed) { killed = rec; code = rcode; }
    }
    if (code) {
        r.status = code;
        r.n_records = std::max<int64_t>(killed, 0);
        r.bytes_consumed = 0; r.lines_consumed = 0;
        if (code == BZQ_BUFFER_EXCEEDED) {
            h->message = "Line exceeds buffer capacity of " + std::to_string(cap) + " bytes";
        } else if (code == BZQ_FASTA_NO_HEADER) {
            int64_t q[3];
            if ((rc = query(h, (int64_t)st.nohdr_pos, q))) return rc;
            r.err_record_number = (int64_t)record_base; r.err_line_number = (int64_t)line_base + q[0] + 1; r.err_file_position = (int64_t)stream_pos + q[1];
            h->message = parse_error_text("FASTA: sequence id line does not start with '>'", r.err_record_number, r.err_line_number, r.err_file_position);
        } else if (code == BZQ_FASTA_EMPTY_SEQUENCE) {
            int64_t hp[2] = {0, 0};
            const bool closed_by_header = killed + 1 < H;
            FACHK(h, hipMemcpy(hp, (const int64_t*)h->hdr_pos.p + killed, closed_by_header ? 16 : 8, hipMemcpyDeviceToHost));
            int64_t q0[3], q1[3] = {0, (int64_t)n, 0};
            if ((rc = query(h, hp[0], q0))) return rc;
            if (closed_by_header && (rc = query(h, hp[1], q1))) return rc;
            r.err_record_number = (int64_t)record_base + killed + 1;
            r.err_line_number = (int64_t)line_base + q0[0] + 2;   // the line after the header line
            r.err_file_position = (int64_t)stream_pos + q1[1];
            h->message = parse_error_text("FASTA record has empty sequence", r.err_record_number, r.err_line_number, r.err_file_position);
        } else {   // ValidationError: the record number is the count so far, 0 is not printed (errors.mojo:223-234)
            r.err_record_number = (int64_t)record_base + killed;
            h->message = "Non ASCII letters found";
            if (r.err_record_number > 0) h->message += "\n  Record number: " + std::to_string(r.err_record_number);
        }
    } else if (is_eof) {
        r.status = BZQ_EOF;
        r.n_records = H;
        r.bytes_consumed = n; r.lines_consumed = nt ? st.lines_consumed : 0;
    } else {
        r.n_records = closed;
        r.status = closed > 0 ? BZQ_OK : BZQ_FASTA_NEED_MORE;
        r.bytes_consumed = nt ? (uint64_t)st.consumed : 0; r.lines_consumed = nt ? st.lines_consumed : 0;
    }
    if (r.n_records > 0) {
        int64_t ends[2];
        FACHK(h, hipMemcpy(&ends[0], (const int64_t*)h->seq_ends.p + (r.n_records - 1), 8, hipMemcpyDeviceToHost));
        FACHK(h, hipMemcpy(&ends[1], (const int64_t*)h->id_ends.p + (r.n_records - 1), 8, hipMemcpyDeviceToHost));
        r.seq_bytes = ends[0]; r.id_bytes = ends[1];
    }
    h->res = r;
    h->killed = code ? killed : INT64_MAX;
    h->n_headers = H;
    *out = r;
    return 0;
}

// internal (bzq_api.hip, bzq_fasta_shard_stitch): what this byte range tells the other ranks -- row = {n, first_header
// (-1 none), lead_kind, tail_open, last_byte} (bzq_fasta.hpp "byte-range shards")
int32_t bzq_fasta_shard_probe_(bzq_fasta* h, const uint8_t* d, uint64_t n, int64_t row[5]) {
    if (!h || (n && !d)) return BZQ_ERR_ARG;
    FACHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->probe, sizeof(ProbeOut)))) return rc;
    ProbeOut po{};
    po.first_header = NONE; po.lead_kind = 3; po.tail_open = -1; po.last_byte = 10;
    if (n > 0) {
        FACHK(h, hipMemcpyAsync(h->probe.p, &po, sizeof po, hipMemcpyHostToDevice, h->stream));
        ProbeArgs a{d, (int64_t)n, (ProbeOut*)h->probe.p, h->cfg.line_capacity};
        const int64_t nt = (int64_t)((n + TILE - 1) / TILE);
        for (int64_t lo = 0, span = 8; lo < nt; lo += span, span *= 8) {
            ProbeHdrArgs ha{d, (int64_t)n, (ProbeOut*)h->probe.p, lo, h->cfg.line_capacity};
            hipLaunchKernelGGL(k_fa_probe_headers, dim3((unsigned)std::min<int64_t>(span, nt - lo)), dim3(BLOCK), 0, h->stream, ha);
        }
        hipLaunchKernelGGL(k_fa_probe_edges, dim3(1), dim3(BLOCK), 0, h->stream, a);
        FACHK(h, hipMemcpyAsync(&po, h->probe.p, sizeof po, hipMemcpyDeviceToHost, h->stream));
        FACHK(h, hipStreamSynchronize(h->stream));
        FACHK(h, hipGetLastError());
    }
    row[0] = (int64_t)n; row[1] = po.first_header == NONE ? -1 : (int64_t)po.first_header; row[2] = po.lead_kind; row[3] = po.tail_open;
    row[4] = po.last_byte;
    return 0;
}

// internal, cold: '\n' count of d[0, n)
int32_t bzq_fasta_count_newlines_(bzq_fasta* h, const uint8_t* d, uint64_t n, int64_t* out) {
    if (!h || !out || (n && !d)) return BZQ_ERR_ARG;
    *out = 0;
    if (!n) return 0;
    FACHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->probe, sizeof(ProbeOut)))) return rc;
    ProbeOut po{};
    FACHK(h, hipMemcpyAsync(h->probe.p, &po, sizeof po, hipMemcpyHostToDevice, h->stream));
    ProbeArgs a{d, (int64_t)n, (ProbeOut*)h->probe.p, 0};
    hipLaunchKernelGGL(k_fa_count_newlines, dim3((unsigned)std::min<uint64_t>((n + BLOCK * 16 - 1) / (BLOCK * 16), 4096)), dim3(BLOCK), 0, h->stream, a);
    FACHK(h, hipMemcpyAsync(&po, h->probe.p, sizeof po, hipMemcpyDeviceToHost, h->stream));
    FACHK(h, hipStreamSynchronize(h->stream));
    *out = po.newlines;
    return 0;
}
int32_t bzq_fasta_shard_scan(bzq_fasta* h, const uint8_t* d_shard, uint64_t n, bzq_fasta_shard_summary* out) {
    if (!out) return BZQ_ERR_ARG;
    int64_t row[5];
    const int32_t rc = bzq_fasta_shard_probe_(h, d_shard, n, row);
    if (rc) return rc;
    *out = bzq_fasta_shard_summary{(uint64_t)row[0], row[1], (int32_t)row[2], (int32_t)row[4], row[3]};
    return 0;
}
int64_t bzq_fasta_error_open_record(const bzq_fasta* h) { return h ? h->killed : INT64_MAX; }
int64_t bzq_fasta_last_headers_(const bzq_fasta* h) { return h ? h->n_headers : 0; }

int32_t bzq_fasta_format_error(bzq_fasta* h, char* buf, size_t cap) {
    if (!h || (cap && !buf)) return BZQ_ERR_ARG;
    if (cap) {
        const size_t k = std::min(cap - 1, h->message.size());
        std::memcpy(buf, h->message.data(), k);
        buf[k] = 0;
    }
    return (int32_t)h->message.size();
}

int32_t bzq_fasta_generate_synthetic_device(bzq_fasta* h, int64_t num_reads, int64_t first, int64_t count, int32_t min_len,
                                            int32_t max_len, int32_t line_width, uint8_t* d_out, uint64_t cap, uint64_t* out_bytes) {
    if (!h || num_reads <= 0 || first < 0 || count < 0 || first + count > num_reads || min_len < 0 || max_len < min_len || line_width <= 0)
        return BZQ_ERR_ARG;
    int nd = 1;
    if (num_reads > 1) nd = (int)std::to_string(num_reads - 1).size();   // utils.mojo:1093-1096
    const int64_t hdr = 6 + nd + 1, range = (int64_t)max_len - min_len + 1;
    auto body = [&](int64_t L) { return L + L / line_width + (L % line_width ? 1 : 0); };
    int64_t period = 1;
    std::vector<int64_t> prefix;
    uint64_t total;
    if (range > 1) {
        period = range / std::gcd<int64_t>(31, range);
        prefix.resize((size_t)period + 1);
        prefix[0] = 0;
        for (int64_t r = 0; r < period; ++r)
            prefix[(size_t)r + 1] = prefix[(size_t)r] + body(min_len + (int64_t)(((uint64_t)r * 31u + 7u) % (uint64_t)range));
        auto sum_to = [&](int64_t i) { return (i / period) * prefix[(size_t)period] + prefix[(size_t)(i % period)]; };
        total = (uint64_t)(count * hdr + sum_to(first + count) - sum_to(first));
    } else {
        total = (uint64_t)((hdr + body(min_len)) * count);
    }
    if (out_bytes) *out_bytes = total;
    if (!d_out) return 0;
    if (cap < total) { h->err = "bzq_fasta_generate_synthetic_device: output buffer too small"; return BZQ_ERR_ARG; }
    FACHK(h, hipSetDevice(h->device));
    int64_t* d_prefix = nullptr;
    if (range > 1) {
        int rc;
        if ((rc = ensure(h, h->gen_prefix, prefix.size() * 8))) return rc;
        d_prefix = (int64_t*)h->gen_prefix.p;
        FACHK(h, hipMemcpyAsync(d_prefix, prefix.data(), prefix.size() * 8, hipMemcpyHostToDevice, h->stream));
    }
    FaGenArgs g{d_out, first, count, min_len, nd, line_width, range, period, d_prefix};
    if (count > 0) hipLaunchKernelGGL(k_fa_generate, dim3((unsigned)((count + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, h->stream, g);
    FACHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

} // extern "C"
