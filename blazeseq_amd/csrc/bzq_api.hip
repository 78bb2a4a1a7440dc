// bzq_api.hip -- host side of libblazeseq_hip.so (C ABI in include/blazeseq_hip.h).
//
// Mirrors, for whole chunks at a time, what blazeseq/fastq/parser.mojo does record by record:
// the ctx is the FastqParser, bzq_submit_*/bzq_chunk_result are next_batch for every batch of the
// chunk at once, bzq_batch_view is the FastqBatch -> DeviceFastqBatch hand-off
// (blazeseq/fastq/record_batch.mojo:89-90, 404-411) without the five allocations, five copies and
// three synchronisations per 4096 records.  There is NO CPU fallback: without a gfx950 device
// bzq_create fails.
#include "../../include/blazeseq_hip.h"
#ifndef BZQ_EXPERIMENTS
#define BZQ_EXPERIMENTS 0
#endif
#if BZQ_EXPERIMENTS
#include "bzq_single.hpp"   // single-launch variants + first-generation kernels: cross-checks and negative results only
#include "bzq_stream.hpp"   // k_stream: one read of the input with super-tiles staged in registers (correct, 2x slower: profiles/r2_single_read.md)
#endif
#include "bzq_views.hpp"
#include "bzq_bufcache.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

using namespace bzq;

namespace {

thread_local std::string g_create_error;

const char* message_for_code(int code) {
    // errors.mojo:71-90
    switch (code) {
        case BZQ_ID_NO_AT: return "Sequence id line does not start with '@'";
        case BZQ_SEP_NO_PLUS: return "Separator line does not start with '+'";
        case BZQ_SEQ_QUAL_LEN_MISMATCH: return "Quality and sequence line do not match in length";
        case BZQ_ASCII_INVALID: return "Non ASCII letters found";
        case BZQ_QUALITY_OUT_OF_RANGE: return "Corrupt quality score according to provided schema";
        case BZQ_UNEXPECTED_EOF: return "Unexpected end of file in FASTQ record";
        case BZQ_BUFFER_EXCEEDED: return "FASTQ record exceeds buffer capacity";
        case BZQ_BUFFER_AT_MAX: return "FASTQ record exceeds maximum buffer capacity";
        default: return "Parse or validation error";
    }
}

// BufferedReader window arithmetic on stream offsets (io/buffered.mojo:137-290), used only to
// decide which terminal error the reference raises for trailing bytes that are not a record.
struct Window {
    int64_t w = 0, end = 0, cap = 0, N = 0;
    bool eof = false;
    int64_t fill() { // _fill_buffer, buffered.mojo:262-281
        if (eof) return 0;
        int64_t space = cap - (end - w);
        if (space == 0) return 0;
        int64_t amt = std::min<int64_t>(space, N - end);
        if (amt < 0) amt = 0;
        end += amt;
        if (amt == 0) eof = true;
        return amt;
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// The device memory one parsed chunk's results live in (see bzq_ctx::out).
struct OutSet {
    DevBuf seq, qual, id;
    DevBuf ends, id_ends, rec_end, b_ends, b_id_ends, off[4], id_start, id_len;
    int64_t rec_cap = 0;
    // rebased ends / id_ends of bzq_batch_view ranges that are not batch aligned: bump-allocated blocks that are never
    // moved or freed while the chunk is live, so every view handed out keeps its own storage
    // ends / id_ends at every batch boundary (k_rebase), mirrored on the host by bzq_chunk_result: an aligned bzq_batch_view
    // then needs no device copy at all
    DevBuf bb;
    int64_t bb_cap = 0;
    int64_t* h_bb = nullptr;       // pinned mirror of bb, filled by an async copy behind k_rebase (arrives with the chunk state)
    int64_t h_bb_cap = 0;          // entries (batches) the mirror can hold
    int64_t h_bb_batches = 0;      // batches copied for the current chunk (0 = not available)
    int64_t h_bb_records = 0;      // complete records the table covers
    // what the chunk that lives in this set needs to have its chunk-cumulative ends derived later (bzq_chunk_cumulative_ends serves
    // the set a bzq_chunk points into: the current one or, under the double-buffer contract, the one before it)
    bool fold = false, cum_valid = false, parsed = false;
    int64_t batch = 0, n_records = 0;
    uint32_t serial = 0;   // ... and which submit it was (bzq_chunk::chunk_serial): bzq_chunk_cumulative_ends refuses a stale bzq_chunk
    std::vector<DevBuf> view_blocks;
    size_t view_used = 0;          // bytes used in view_blocks.back()
    size_t view_next = 4u << 20;   // size of the next block
};

} // namespace

struct bzq_ctx {
    int device = 0;
    bzq_config cfg{};
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // emit kernels of sub-chunk k overlap pass A of sub-chunk k+1
    std::vector<hipEvent_t> ev_pipe;
    int overlap = 0;
    bool own_stream = true;
    std::string err;
    // arenas.  Everything a bzq_chunk / bzq_device_batch / bzq_device_views points at lives in an OutSet; there are two
    // of them and submit k writes set k & 1, so a chunk's results stay valid until the SECOND following bzq_submit_* (a
    // consumer kernel on another stream may still read chunk k while chunk k+1 is parsed).  Set 1 is allocated by the
    // first submit that needs it.
    DevBuf in;
    OutSet out[2];
    int cur_set = 0;
    OutSet& o() { return out[cur_set]; }
    const OutSet& o() const { return out[cur_set]; }
    int64_t n_submits = 0;
    int double_buffer = 1;   // option "double_buffer": 0 = one set (results valid until the next submit), half the memory
    hipStream_t consumer_stream = nullptr;   // bzq_set_consumer_stream: where the bzq_batch_* / bzq_column_* kernels run (default: the ctx stream)
    DevBuf tile_c, tile_a, tile_idc, tileP, tileS, tileQ, tileI, grp, desc, consumer_scratch, qpos_scratch, gen_prefix, entries, tile_list, tile_vf, inflate_tab;
    int64_t tile_cap = 0;
    ChunkState* d_state = nullptr;
    ViewsPool* d_pool = nullptr;   // views mode: the entry pool's ticket (bzq_views.hpp), zero between chunks
    ChunkState* h_state = nullptr; // pinned
    hipEvent_t ev[8]{};
    std::vector<hipEvent_t> ev_detail;        // events recorded by the current submit (option timing_detail) ...
    std::vector<hipEvent_t> ev_detail_pool;   // ... taken from here: created once (creating and destroying eight events per submit was 2 % of a step)
    size_t ev_detail_used = 0;
    // options
    int ablate = 0;
    int64_t pool_slots = 0;
    bool views_bytes_once = false;   // this chunk is being repeated on the byte-level kernels (pool ran out)
    int views_bytes = 0;   // option: views mode through the two-read kernels even without validation (cross-check)
    int force_dense = 0, timing_detail = 0, single_pass = 0, v2 = 1, num_cu = 256;
    bool ran_single_pass = false;
    int inflate_ms = 0;                       // option "inflate_ms": 1 = BGZF blocks eight to a wave (bzq_inflate_ms.hpp: correct, measured SLOWER -- profiles/r4_inflate_ms.md); 0 = one block per wave (bzq_inflate.hpp)
    int ingest_gpu_inflate = 1;               // option "ingest_gpu_inflate": BGZF blocks are inflated on the device (bzq_inflate.hpp), 0 = on the reader threads
    int ingest_direct = 0, ingest_numa = 1;   // options "ingest_direct" (O_DIRECT reads), "ingest_numa" (bind readers to the GPU's node)
    // option "fold_rebase" (default 1): the emit kernel writes the per-batch ends and does the record-length check itself, from the
    // batch bases k_batch_bases computes between the scan and the emit (FusedArgs::fold) -- no pass over the per-record arrays
    int fold_opt = 1;
    bool fold = false;         // ... decided per chunk (decide_fold)
    bool cum_valid = false;    // the current chunk's chunk-cumulative ends / id_ends hold values (always without fold; with it: after bzq_chunk_cumulative_ends)
    bool finish_done = false;  // k_tail of this submit already left the chunk totals (enqueue_rebase skips k_finish once)
    int stream_prio = 0; bool stream_has_prio = false;   // the priority `stream` was created with (stream_init gets the same)
    bool published = false;    // ... and wrote the chunk state and the batch table into the host's pinned copies itself (no copy packets behind the kernels)
    // the state's initial values travel on a stream of their own while pass A runs (it does not look at the state); the scan waits for them
    hipStream_t stream_init = nullptr;
    hipEvent_t ev_init = nullptr;
    bool pool_dirty = false;      // views mode: pass A may have taken pool tickets that no scan has folded and zeroed yet (a submit that failed in between)
    bool init_in_flight = false;
    bool init_deferred = false;   // ... and are handed to the side stream only BEHIND pass A's launch (the host's copy call is not in front of the first kernel)
    int init_in_kernel = 1;    // option "state_init_in_kernel": 1 (default; 2-4: variants of the race hunt) = ScanArgs::init_mode, 0 = copied in on a side stream
    bool init_stream_failed = false;
    bool init_by_kernel_now = false;
    int64_t h_init[4] = {0, 0, 0, 0};
    int lean = 1;              // option "lean_submit": 0 = state copies on the ctx stream, before and behind the kernels (as before)
    DevBuf tile_last, tileB, btile;
    // bzq_shard_read_range: the rank's byte range of a file in device memory, and the pinned pieces it travelled through
    DevBuf shard_buf;
    std::vector<void*> shard_pin;          // 2 per reader thread, SHARD_PIECE bytes each (pinned once, reused)
    std::vector<hipStream_t> shard_streams;
    std::vector<hipEvent_t> shard_events;
    int pass_a_h = 1;          // option "pass_a_h": pass A from the newline bitmap alone (k_tile_aggregate_h), verified by the emit
    bool exact_pass_a = false; // this chunk is being repeated with the exact pass A
    bool used_h = false;
    // The hypothesis of k_tile_aggregate_h ("no id loses bytes to _strip_spaces") is a property of the FILE, not of a chunk: a CRLF
    // file strips a '\r' from every id, and trying the hypothesis on each of its chunks parses every chunk twice.  After a
    // contradiction the following exact_sticky submits go straight to the exact pass A (16, then doubling up to 1024 each time
    // the retried hypothesis fails again); option "pass_a_sticky" = 0 restores the per-chunk retry.
    int sticky_opt = 1;
    int exact_sticky = 0, exact_sticky_len = 0;
    int64_t hyp_contradictions = 0;
    int use_stream = 0;        // option "stream" (EXPERIMENTS build): batch mode through the single-read kernel k_stream
    bool ran_stream = false;
    int64_t stream_fallbacks = 0;
    // current chunk
    const uint8_t* cur = nullptr;
    uint64_t cur_n = 0, cur_stream_pos = 0;
    int cur_is_eof = 0;
    uint32_t cur_prev_byte = 10;
    int64_t cur_first_header = 0;
    // option "consumer_guard" (0 default): 1 = the library itself keeps the two-chunk lifetime rule for work on the consumer stream.
    // Every submit records an event on the consumer stream and waits -- on the device, the host does not block -- for the event the
    // PREVIOUS submit recorded: what the consumer stream held when chunk k was submitted (the consumers of chunk k - 1 and before,
    // for a host that enqueues a chunk's consumers before it submits the next chunk) is through before chunk k + 1's kernels start
    // writing chunk k - 1's output set, and chunk k + 1's parse still overlaps chunk k's consumers (bench.py pipeline_mode)
    int consumer_guard = 0;
    hipEvent_t ev_guard[2] = {nullptr, nullptr};
    int64_t guard_serial = 0;   // submits that recorded an event
    int64_t dbg[3][12] = {};        // diagnostic: see bzq_chunk_result (query "dump_state")
    int64_t last_dense_tiles = 0;   // query "dense_tiles": tiles of the last parsed chunk that took the serial path (diagnostic)
    bool pending = false, have_result = false;
    int64_t n_passes = 0;
    bzq_chunk res{};
    // The chunk `res` describes lives in out[res_set] and stays there until the SECOND submit after its own (the double-buffer
    // contract): bzq_batch_view / bzq_batches / bzq_views keep serving it while the NEXT chunk is already being parsed, so a host can
    // take a result, submit the next chunk at once and walk the batches of the one it holds under the parse of the next.
    int res_set = 0;
    bool res_alive = false;
    const uint8_t* res_cur = nullptr;   // the device chunk `res` belongs to (bzq_device_views::chunk)
    // terminal-status details for bzq_format_error
    int term_phase = 0;
    int64_t term_cap = 0;
    // shard mode
    bool shard_mode = false;
    const uint8_t* agg_ptr = nullptr;  // shard whose tile aggregates are already in the arenas
    uint64_t agg_n = 0;
    int head_lines = 0;
    // last shard's trailing bytes (see bzq_chunk_result): 0 decide locally, 1 defer to the protocol, 2 decided by it
    int tail_mode = 0;
    bool tail_pending = false, tail_accept = false;
    int tail_code = 0, tail_phase_dec = 0;
    int64_t tail_cap_dec = 0;
    struct bzq_comm* comm = nullptr;   // multi-GPU exchange (bzq_comm.hpp), owned
    int64_t comm_timeout_ms = 120000;  // option "comm_timeout_ms": host-side deadline of every exchange of the shard protocols
    uint64_t shard_totals[3] = {0, 0, 0};   // records, bases, bytes over all ranks after the last bzq_shard_stitch
    bool have_shard_totals = false;
    int ranks_seen = 0;                // rows of the last bzq_shard_stitch's summary all-gather that carried their rank's stamp (query "ranks_seen")
    float ms_scan_shard = 0.f;         // kernels of the bzq_shard_scan that preceded this submit (pass A + scan)
    // A stream parsed in several chunks: where the reference's BufferedReader window stands behind every record handed out so
    // far (option "records_before").  Which outcome the reference gives for a stream that ends in bytes that are not a record
    // depends on that window, and the window on every record since the stream's first byte -- but only through
    // a state of three integers.  The record ends of chunk k come back to the host behind its parse (8 B per record, pinned,
    // asynchronous) and are walked while chunk k + 1 is being parsed, as soon as the caller has said how many of them it took
    // (records_before of the next submit).  (Rounds 1-2 kept every record end of the whole stream on the device for this:
    // 5 GB for a 625 M-read stream.)
    int64_t records_before = -1;   // option "records_before": records delivered by earlier chunks; < 0 = every chunk a stream of its own
    Window follow;                 // the window behind the last record walked (N = "unbounded" until the stream's end is known)
    int64_t follow_head = 0, follow_records = 0;   // stream offset behind that record; records walked
    bool follow_on = false;
    int64_t* stage_pin = nullptr;  // record ends of the last parsed chunk, chunk relative (pinned)
    size_t stage_cap = 0;
    int64_t stage_n = 0, stage_rb = 0, stage_pos = 0, stage_head = 0;
    bool stage_valid = false;
    hipEvent_t stage_ev = nullptr;
};

namespace {

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return BZQ_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

// option timing_detail: an event on the ctx stream between two phases of the current submit
// (every event between two kernels costs 3-6 us of an idle queue -- scripts/probes/event_cost_probe.hip, scripts/step_gaps.py -- so
// a mark that would sit right behind another event takes that event instead: `same_as`)
void mark_detail(bzq_ctx* c, hipEvent_t same_as = nullptr) {
    if (same_as) { c->ev_detail.push_back(same_as); return; }
    if (c->ev_detail_used == c->ev_detail_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->ev_detail_pool.push_back(e);
    }
    hipEvent_t e = c->ev_detail_pool[c->ev_detail_used++];
    (void)hipEventRecord(e, c->stream);
    c->ev_detail.push_back(e);
}

int ensure(bzq_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {   // the library's own buffer cache may be what is in the way (bzq_bufcache.hpp): give it back, once
        (void)hipGetLastError();
        bzq::cache::device_pool().trim(0, true);
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        c->err = "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e);
        b.p = nullptr;
        return BZQ_ERR_NOMEM;
    }
    b.cap = bytes;
    return 0;
}

constexpr int64_t BB_MAX_BATCHES = 1ll << 20;   // batch-boundary table kept for chunks of at most this many batches (16 MiB)
int follow_consume(bzq_ctx* c);   // (defined behind the window arithmetic)

int64_t tiles_for(uint64_t n) { return (int64_t)((n + TILE - 1) / TILE); }

// per-tile summaries and prefixes (shared by both output sets: they only live between the kernels of one chunk)
int ensure_tile_arenas(bzq_ctx* c, uint64_t n) {
    int rc;
    const int64_t nt = tiles_for(n) + 1;
    if (nt > c->tile_cap) {
        if ((rc = ensure(c, c->tile_c, nt * 4)) || (rc = ensure(c, c->tile_a, nt * 8)) ||
            (rc = ensure(c, c->tile_idc, nt * 8)) || (rc = ensure(c, c->tileP, nt * 8)) ||
            (rc = ensure(c, c->tileS, nt * 8)) || (rc = ensure(c, c->tileQ, nt * 8)) ||
            (rc = ensure(c, c->tileI, nt * 8)) || (rc = ensure(c, c->grp, (nt / SG_TILES + 2) * 80)) ||
            (rc = ensure(c, c->tile_last, nt * 8)) || (rc = ensure(c, c->tileB, nt * 4)))
            return rc;
#if BZQ_EXPERIMENTS
        if ((rc = ensure(c, c->desc, (nt * 6 + (nt / 64 + 2) * 4 + 64) * 8))) return rc;   // single-launch variants only
#endif
        c->tile_cap = nt;
    }
    if (c->cfg.views_only) {   // line entries: a 4 KiB slot per tile + a pool of 64 KiB slots for tiles of tiny records
        c->pool_slots = std::max<int64_t>(16, nt / 8);
        if ((rc = ensure(c, c->entries, (size_t)nt * ENT_STRIDE * 4)) || (rc = ensure(c, c->tile_list, (size_t)c->pool_slots * TILE * 4)) ||
            (rc = ensure(c, c->tile_vf, (size_t)nt + 64)))
            return rc;
    }
    return 0;
}

// the three columns of the current output set
int ensure_col_arenas(bzq_ctx* c, uint64_t n) {
    if (c->cfg.views_only) return 0;   // views mode packs nothing
    int rc;
    const size_t col = (size_t)n + 64;
    if ((rc = ensure(c, c->o().seq, col)) || (rc = ensure(c, c->o().qual, col)) || (rc = ensure(c, c->o().id, col))) return rc;
    return 0;
}

// A new chunk is about to be parsed: move to the other output set (its previous contents -- the chunk before the
// previous one -- are no longer valid) and recycle its view storage.
int begin_submit(bzq_ctx* c) {
    if (c->pending) HIPCHK(c, hipStreamSynchronize(c->stream)); // h_state is reused
    if (c->double_buffer) c->cur_set = (int)(c->n_submits & 1);
    c->n_submits += 1;
    if (!c->double_buffer || c->cur_set == c->res_set) c->res_alive = false;   // the set the last result lives in is written again
    OutSet& o = c->o();
    if (o.view_blocks.size() > 1) {   // settle on one block as large as everything the last chunk needed
        size_t total = 0;
        for (DevBuf& b : o.view_blocks) { total += b.cap; if (b.p) HIPCHK(c, hipFree(b.p)); }
        o.view_blocks.clear();
        o.view_next = std::max(o.view_next, total);
    }
    o.view_used = 0;
    return 0;
}

// storage for the rebased ends of one unaligned batch view: stays where it is until the set is recycled
int view_alloc(bzq_ctx* c, OutSet& o, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (o.view_blocks.empty() || o.view_used + bytes > o.view_blocks.back().cap) {
        DevBuf b;
        int rc;
        if ((rc = ensure(c, b, std::max(bytes, o.view_next)))) return rc;
        o.view_blocks.push_back(b);
        o.view_used = 0;
    }
    *out = (uint8_t*)o.view_blocks.back().p + o.view_used;
    o.view_used += bytes;
    return 0;
}

int ensure_record_arenas(bzq_ctx* c, int64_t recs) {
    if (recs <= c->o().rec_cap) return 0;
    int rc;
    const size_t b = (size_t)recs * 8;
    if ((rc = ensure(c, c->o().ends, b)) || (rc = ensure(c, c->o().id_ends, b)) || (rc = ensure(c, c->o().rec_end, b)) ||
        (rc = ensure(c, c->o().b_ends, b)) || (rc = ensure(c, c->o().b_id_ends, b)))
        return rc;
    if (c->cfg.emit_offsets || c->cfg.views_only)
        for (int i = 0; i < 4; ++i)
            if ((rc = ensure(c, c->o().off[i], b))) return rc;
    if (c->cfg.views_only && ((rc = ensure(c, c->o().id_start, b)) || (rc = ensure(c, c->o().id_len, b / 2)))) return rc;
    c->o().rec_cap = recs;
    return 0;
}


int64_t pass_tiles(const bzq_ctx* c) {
    int64_t pb = c->cfg.pass_bytes > 0 ? c->cfg.pass_bytes : (int64_t)1 << 40;
    int64_t t = pb / TILE;
    return t < 1 ? 1 : t;
}

#if BZQ_EXPERIMENTS
template <bool CA, bool CQ>
void launch_emit_off(bool offs, dim3 grid, hipStream_t s, const EmitArgs& a) {
    if (offs) hipLaunchKernelGGL((k_tile_emit<CA, CQ, true>), grid, dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL((k_tile_emit<CA, CQ, false>), grid, dim3(BLOCK), 0, s, a);
}
void launch_emit(const bzq_ctx* c, dim3 grid, const EmitArgs& a) {
    const bool ca = c->cfg.check_ascii != 0, cq = c->cfg.check_quality != 0, off = c->cfg.emit_offsets != 0;
    if (ca && cq) launch_emit_off<true, true>(off, grid, c->stream, a);
    else if (ca) launch_emit_off<true, false>(off, grid, c->stream, a);
    else if (cq) launch_emit_off<false, true>(off, grid, c->stream, a);
    else launch_emit_off<false, false>(off, grid, c->stream, a);
}

EmitArgs make_emit_args(bzq_ctx* c) {
    EmitArgs e{};
    e.g = c->cur; e.n = (int64_t)c->cur_n; e.prev_byte = c->cur_prev_byte;
    e.tileP = (const int64_t*)c->tileP.p; e.tileS = (const int64_t*)c->tileS.p;
    e.tileQ = (const int64_t*)c->tileQ.p; e.tileI = (const int64_t*)c->tileI.p;
    e.col_seq = (uint8_t*)c->o().seq.p; e.col_qual = (uint8_t*)c->o().qual.p; e.col_id = (uint8_t*)c->o().id.p;
    e.ends = (int64_t*)c->o().ends.p; e.id_ends = (int64_t*)c->o().id_ends.p; e.rec_end = (int64_t*)c->o().rec_end.p;
    e.rec_cap = c->o().rec_cap;
    e.o_hdr = (int64_t*)c->o().off[0].p; e.o_seq = (int64_t*)c->o().off[1].p;
    e.o_sep = (int64_t*)c->o().off[2].p; e.o_qual = (int64_t*)c->o().off[3].p;
    e.st = c->d_state; e.q_lower = c->cfg.q_lower; e.q_upper = c->cfg.q_upper;
    e.force_dense = c->force_dense;
    return e;
}

#endif

template <bool CA, bool CQ, bool OFFS, bool LB>
void launch_fused_one(const bzq_ctx* c, dim3 grid, const FusedArgs& a) {
    if constexpr (!LB) {
        if (a.fold) { hipLaunchKernelGGL((k_fused<CA, CQ, OFFS, false, true>), grid, dim3(BLOCK), 0, c->stream, a); return; }
    }
    hipLaunchKernelGGL((k_fused<CA, CQ, OFFS, LB>), grid, dim3(BLOCK), 0, c->stream, a);
}
template <bool CA, bool CQ, bool LB>
void launch_fused_off(const bzq_ctx* c, bool offs, dim3 grid, const FusedArgs& a) {
    if (offs) launch_fused_one<CA, CQ, true, LB>(c, grid, a);
    else launch_fused_one<CA, CQ, false, LB>(c, grid, a);
}
template <bool LB>
void launch_fused(const bzq_ctx* c, dim3 grid, const FusedArgs& f) {
    const bool ca = c->cfg.check_ascii != 0, cq = c->cfg.check_quality != 0, off = c->cfg.emit_offsets != 0;
    if (ca && cq) launch_fused_off<true, true, LB>(c, off, grid, f);
    else if (ca) launch_fused_off<true, false, LB>(c, off, grid, f);
    else if (cq) launch_fused_off<false, true, LB>(c, off, grid, f);
    else launch_fused_off<false, false, LB>(c, off, grid, f);
}
// how far header_kept may walk outside its tile: the longest record the reference's buffer can hold
int64_t walk_limit_of(const bzq_ctx* c) { return c->cfg.buffer_growth_enabled ? c->cfg.buffer_max_capacity : c->cfg.buffer_capacity; }

FusedArgs make_fused_args(bzq_ctx* c) {
    FusedArgs f{};
    f.g = c->cur; f.n = (int64_t)c->cur_n; f.prev_byte = c->cur_prev_byte; f.n_tiles = tiles_for(c->cur_n);
    f.tileP = (const int64_t*)c->tileP.p; f.tileS = (const int64_t*)c->tileS.p;
    f.tileQ = (const int64_t*)c->tileQ.p; f.tileI = (const int64_t*)c->tileI.p;
    f.col_seq = (uint8_t*)c->o().seq.p; f.col_qual = (uint8_t*)c->o().qual.p; f.col_id = (uint8_t*)c->o().id.p;
    f.ends = (int64_t*)c->o().ends.p; f.id_ends = (int64_t*)c->o().id_ends.p; f.rec_end = (int64_t*)c->o().rec_end.p;
    f.rec_cap = c->o().rec_cap;
    f.o_hdr = (int64_t*)c->o().off[0].p; f.o_seq = (int64_t*)c->o().off[1].p;
    f.o_sep = (int64_t*)c->o().off[2].p; f.o_qual = (int64_t*)c->o().off[3].p;
    f.st = c->d_state; f.q_lower = c->cfg.q_lower; f.q_upper = c->cfg.q_upper; f.force_dense = c->force_dense; f.ablate = c->ablate;
    f.walk_limit = walk_limit_of(c);
    f.check_h = c->used_h ? 1 : 0;
    if (c->fold) {
        f.fold = 1;
        f.b_ends = (int64_t*)c->o().b_ends.p; f.b_id_ends = (int64_t*)c->o().b_id_ends.p;
        f.bb = (const int64_t*)c->o().bb.p; f.bb_cap = c->o().bb_cap;
        f.tileB = (const int32_t*)c->tileB.p; f.tile_last = (const u64*)c->tile_last.p;
        f.batch = c->cfg.batch_size; f.first_header = c->cur_first_header;
        f.len_limit = c->cfg.buffer_growth_enabled ? c->cfg.buffer_max_capacity : c->cfg.buffer_capacity;
    }
    return f;
}

// The batch-boundary table of the current output set (bzq_batch_view's host cache; with fold also the emit's batch bases): only
// while it stays small -- batch sizes of a few records on a huge chunk fall back to per-view copies.
void ensure_bb(bzq_ctx* c) {
    const int64_t nb_max = c->o().rec_cap / std::max<int64_t>(1, c->cfg.batch_size) + 2;
    int64_t* bb = nullptr;
    if (nb_max <= BB_MAX_BATCHES && ensure(c, c->o().bb, (size_t)nb_max * 16) == 0) bb = (int64_t*)c->o().bb.p;
    c->o().bb_cap = bb ? nb_max : 0;
}

// Chunk-cumulative ends / id_ends at record r of the current chunk, on the host (cold paths: a stream cut short by an error, a
// batch view that is not batch aligned).  With fold the emit wrote only the per-batch arrays: value + the batch's base.
int cum_pair(bzq_ctx* c, const OutSet& o, bool fold, bool cum_valid, int64_t r, int64_t out[2]) {
    if (!fold || cum_valid) {
        HIPCHK(c, hipMemcpy(&out[0], (const int64_t*)o.ends.p + r, 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&out[1], (const int64_t*)o.id_ends.p + r, 8, hipMemcpyDeviceToHost));
        return 0;
    }
    const int64_t k = r / std::max<int64_t>(1, c->cfg.batch_size);
    int64_t base[2] = {0, 0};
    HIPCHK(c, hipMemcpy(&out[0], (const int64_t*)o.b_ends.p + r, 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&out[1], (const int64_t*)o.b_id_ends.p + r, 8, hipMemcpyDeviceToHost));
    if (k > 0) HIPCHK(c, hipMemcpy(base, (const int64_t*)o.bb.p + 2 * (k - 1), 16, hipMemcpyDeviceToHost));
    out[0] += base[0]; out[1] += base[1];
    return 0;
}
int cum_pair(bzq_ctx* c, int64_t r, int64_t out[2]) { return cum_pair(c, c->o(), c->fold, c->cum_valid, r, out); }

// Can this chunk go without k_rebase?  One pass over the chunk (the bases of a later sub-chunk pass are not known when an
// earlier one is emitted), at most one batch boundary per fast-path tile, no host-SIMD-width quirk of the quality check
// (that one reads the quality bytes per record), and the table fits.
void decide_fold(bzq_ctx* c) {
    c->fold = false;
    if (!c->fold_opt || !c->v2 || c->cfg.views_only || c->ran_single_pass || c->ran_stream || c->overlap || c->cur_n == 0) return;
    if (tiles_for(c->cur_n) > pass_tiles(c)) return;
    if ((int64_t)c->cfg.batch_size < FOLD_MIN_BATCH) return;
    if (c->cfg.check_quality && c->cfg.compat_simd_width != 0) return;
    ensure_bb(c);
    c->fold = c->o().bb_cap > 0 && ensure(c, c->btile, (size_t)c->o().bb_cap * 8) == 0;
}

// the per-record outputs of an accepted unterminated last record (parser.mojo:464-475): the columns already hold its bytes
void launch_fix_last(bzq_ctx* c, int64_t rec, int64_t n, int64_t batch) {
    if (c->fold && !c->cum_valid)
        hipLaunchKernelGGL(k_fix_last_b, dim3(1), dim3(64), 0, c->stream, rec, n, batch, (int64_t*)c->o().rec_end.p, (int64_t*)c->o().b_ends.p,
                           (int64_t*)c->o().b_id_ends.p, (int64_t*)c->o().bb.p, c->o().bb_cap, (const ChunkState*)c->d_state);
    else
        hipLaunchKernelGGL(k_fix_last, dim3(1), dim3(64), 0, c->stream, rec, n, batch, (int64_t*)c->o().ends.p, (int64_t*)c->o().id_ends.p,
                           (int64_t*)c->o().rec_end.p, (int64_t*)c->o().b_ends.p, (int64_t*)c->o().b_id_ends.p, (const ChunkState*)c->d_state);
}

ChunkFinishArgs finish_args(bzq_ctx* c, bool on) {
    ChunkFinishArgs f{};
    if (on && c->fold) {
        f.on = 1;
        f.b_ends = (const int64_t*)c->o().b_ends.p; f.b_id_ends = (const int64_t*)c->o().b_id_ends.p; f.rec_end = (const int64_t*)c->o().rec_end.p;
        f.batch = std::max<int64_t>(1, c->cfg.batch_size); f.first_header = c->cur_first_header; f.rec_cap = c->o().rec_cap;
        f.bb = c->o().bb_cap ? (int64_t*)c->o().bb.p : nullptr; f.bb_cap = c->o().bb_cap;
        if (c->published) { f.h_state = c->h_state; f.h_bb = c->o().h_bb; f.h_bb_cap = c->o().h_bb_cap; }
    }
    return f;
}

// the pinned mirror of the batch-boundary table (arrives with the chunk state)
void ensure_h_bb(bzq_ctx* c) {
    OutSet& o = c->o();
    const int64_t nb_max = o.rec_cap / std::max<int64_t>(1, c->cfg.batch_size) + 2;
    if (o.bb_cap && o.h_bb_cap < nb_max) {
        if (o.h_bb) (void)hipHostFree(o.h_bb);
        o.h_bb = nullptr; o.h_bb_cap = 0;
        if (hipHostMalloc((void**)&o.h_bb, (size_t)nb_max * 16, hipHostMallocDefault) == hipSuccess) o.h_bb_cap = nb_max;
        else (void)hipGetLastError();
    }
}

void launch_batch_bases(bzq_ctx* c) {
    BasesArgs b{c->cur, (int64_t)c->cur_n, c->cur_prev_byte, tiles_for(c->cur_n), (const int64_t*)c->tileP.p, (const int64_t*)c->tileQ.p,
                (const int64_t*)c->tileI.p, (int64_t)c->cfg.batch_size, (int64_t*)c->o().bb.p, c->o().bb_cap, c->d_state, walk_limit_of(c), (const int64_t*)c->btile.p};
    // boundaries that can exist: the chunk holds at most n / 4 records of four newlines
    const int64_t nb = std::min<int64_t>(c->o().bb_cap, (int64_t)(c->cur_n / 4) / std::max<int64_t>(1, c->cfg.batch_size));
    if (nb > 0) hipLaunchKernelGGL(k_batch_bases, dim3((unsigned)nb), dim3(BLOCK), 0, c->stream, b);
}

#if BZQ_EXPERIMENTS
// One launch for the whole chunk (single-pass kernel, bzq_fused.hpp).
static __global__ void k_pick_run_bases(const int64_t* P, const int64_t* S, const int64_t* Q, const int64_t* I, int64_t run, int64_t nt, int64_t* out) {
    const int e = threadIdx.x;
    const int64_t t = (int64_t)e * run;
    if (t < nt) { out[4 * e] = P[t]; out[4 * e + 1] = S[t]; out[4 * e + 2] = Q[t]; out[4 * e + 3] = I[t]; }
}
int enqueue_fused(bzq_ctx* c) {
    const int64_t nt = tiles_for(c->cur_n);
    u64* d = (u64*)c->desc.p;
    hipError_t e = hipMemsetAsync(d, 0, (size_t)(nt * 5 + 8 + 32) * 8, c->stream);
    if (e != hipSuccess) { c->err = std::string("hipMemsetAsync(desc): ") + hipGetErrorString(e); return BZQ_ERR_HIP; }
    FusedArgs f = make_fused_args(c);
    f.ticket = d; f.desc_c = d + 8; f.desc_agg = d + 8 + nt; f.desc_pre = d + 8 + 2 * nt;
    if (c->single_pass == 4) {   // round-3 experiment: one look-back chain per XCD, the chain starts given (tile prefixes of the two-pass run before)
        f.xcd_tiles = (nt + 7) / 8;
        int64_t* xb = (int64_t*)(d + 8 + 5 * nt);
        hipLaunchKernelGGL(k_pick_run_bases, dim3(1), dim3(8), 0, c->stream, (const int64_t*)c->tileP.p, (const int64_t*)c->tileS.p, (const int64_t*)c->tileQ.p,
                           (const int64_t*)c->tileI.p, f.xcd_tiles, nt, xb);
        f.xcd_base = xb;
    }
    if (c->timing_detail) mark_detail(c);
    launch_fused<true>(c, dim3((unsigned)nt), f);
    if (c->timing_detail) mark_detail(c);
    c->n_passes = 1;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("kernel launch: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    return 0;
}

template <bool CA, bool CQ>
void launch_single_off(const bzq_ctx* c, bool offs, int mode, dim3 grid, const SingleArgs& a) {
    if (mode == 0) {
        if (offs) hipLaunchKernelGGL((k_single<CA, CQ, true, 0>), grid, dim3(BLOCK), 0, c->stream, a);
        else hipLaunchKernelGGL((k_single<CA, CQ, false, 0>), grid, dim3(BLOCK), 0, c->stream, a);
    } else {
        if (offs) hipLaunchKernelGGL((k_single<CA, CQ, true, 1>), grid, dim3(BLOCK), 0, c->stream, a);
        else hipLaunchKernelGGL((k_single<CA, CQ, false, 1>), grid, dim3(BLOCK), 0, c->stream, a);
    }
}

// One launch for the whole chunk: tiles + one prefix-service workgroup (bzq_single.hpp).
int enqueue_single(bzq_ctx* c) {
    const int64_t nt = tiles_for(c->cur_n);
    u64* d = (u64*)c->desc.p;
    const int64_t nb = nt / HB + 2;
    hipError_t e = hipMemsetAsync(d, 0, (size_t)(nt * 6 + nb * 4 + 16) * 8, c->stream);
    if (e != hipSuccess) { c->err = std::string("hipMemsetAsync(desc): ") + hipGetErrorString(e); return BZQ_ERR_HIP; }
    SingleArgs sa{};
    sa.f = make_fused_args(c);
    sa.f.ticket = d;
    sa.dc = d + 16; sa.pc = sa.dc + nt; sa.da = sa.pc + nt; sa.ps = sa.da + nt; sa.pq = sa.ps + nt; sa.pi = sa.pq + nt;
    sa.bc = sa.pi + nt; sa.bs = sa.bc + nb; sa.bq = sa.bs + nb; sa.bi = sa.bq + nb;
    const int mode = c->single_pass == 3 ? 1 : 0;
    if (c->timing_detail) mark_detail(c);
    const dim3 grid((unsigned)(nt + (mode == 0 ? 1 : 0)));
    const bool ca = c->cfg.check_ascii != 0, cq = c->cfg.check_quality != 0, off = c->cfg.emit_offsets != 0;
    if (ca && cq) launch_single_off<true, true>(c, off, mode, grid, sa);
    else if (ca) launch_single_off<true, false>(c, off, mode, grid, sa);
    else if (cq) launch_single_off<false, true>(c, off, mode, grid, sa);
    else launch_single_off<false, false>(c, off, mode, grid, sa);
    if (c->timing_detail) mark_detail(c);
    c->n_passes = 1;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("kernel launch: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    return 0;
}

#endif

#if BZQ_EXPERIMENTS
// Batch mode with one read of the input: one launch of k_stream (bzq_stream.hpp) for the whole chunk.
template <bool CA, bool CQ>
void launch_stream_off(const bzq_ctx* c, bool offs, dim3 grid, const StreamArgs& a) {
    if (offs) hipLaunchKernelGGL((k_stream<CA, CQ, true>), grid, dim3(BLOCK), 0, c->stream, a);
    else hipLaunchKernelGGL((k_stream<CA, CQ, false>), grid, dim3(BLOCK), 0, c->stream, a);
}
int enqueue_stream(bzq_ctx* c) {
    const int64_t nt = tiles_for(c->cur_n);
    const int64_t n_wg = (nt + ST - 1) / ST, n_grp = (n_wg + SGRP - 1) / SGRP;
    const size_t words = (size_t)n_wg * WD_WORDS + (size_t)n_grp * GD_WORDS + 8;
    int rc;
    if ((rc = ensure(c, c->desc, words * 8))) return rc;
    HIPCHK(c, hipMemsetAsync(c->desc.p, 0, words * 8, c->stream));
    StreamArgs sa{};
    sa.f = make_fused_args(c);
    sa.wd = (u64*)c->desc.p; sa.gd = sa.wd + (size_t)n_wg * WD_WORDS; sa.ticket = sa.gd + (size_t)n_grp * GD_WORDS; sa.n_wg = n_wg;
    if (c->timing_detail) mark_detail(c);
    const dim3 grid((unsigned)n_wg);
    const bool ca = c->cfg.check_ascii != 0, cq = c->cfg.check_quality != 0, off = c->cfg.emit_offsets != 0;
    if (ca && cq) launch_stream_off<true, true>(c, off, grid, sa);
    else if (ca) launch_stream_off<true, false>(c, off, grid, sa);
    else if (cq) launch_stream_off<false, true>(c, off, grid, sa);
    else launch_stream_off<false, false>(c, off, grid, sa);
    if (c->timing_detail) mark_detail(c);
    c->n_passes = 1;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("kernel launch: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    return 0;
}

#else
int enqueue_stream(bzq_ctx* c) { c->err = "k_stream exists only in an EXPERIMENTS build"; return BZQ_ERR_ARG; }
#endif

int enqueue_single_launch(bzq_ctx* c) {
#if BZQ_EXPERIMENTS
    return (c->single_pass == 2 || c->single_pass == 3) ? enqueue_single(c) : enqueue_fused(c);   // 1: one look-back chain, 4: one per XCD (experiment), 2 / 3: service / hierarchical
#else
    c->err = "single-launch variants exist only in an EXPERIMENTS build";
    return BZQ_ERR_ARG;
#endif
}

bool views_meta(const bzq_ctx* c);
void launch_scan(bzq_ctx* c, int64_t tb, int64_t te, int pass) {
    ScanArgs s{tb, te, (const uint32_t*)c->tile_c.p, (const u64*)c->tile_a.p, (const u64*)c->tile_idc.p,
               (int64_t*)c->tileP.p, (int64_t*)c->tileS.p, (int64_t*)c->tileQ.p, (int64_t*)c->tileI.p,
               (int64_t*)c->grp.p, c->d_state, pass, c->fold ? (int32_t*)c->tileB.p : nullptr, std::max<int64_t>(1, c->cfg.batch_size),
               c->fold ? (int64_t*)c->btile.p : nullptr, c->fold ? c->o().bb_cap : 0, views_meta(c) ? c->d_pool : nullptr,
               c->init_by_kernel_now ? c->init_in_kernel : 0, {c->h_init[0], c->h_init[1], c->h_init[2], c->h_init[3]}};
    c->init_by_kernel_now = false;   // (only the first scan of a submit: a repeat keeps what the host copied in)
    const int64_t ng = (te - tb + SG_TILES - 1) / SG_TILES;
    if (ng <= 0) return;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)ng), dim3(SG_THREADS), 0, c->stream, s);
    hipLaunchKernelGGL(k_scan_down, dim3((unsigned)ng), dim3(SG_THREADS), 0, c->stream, s);
    if (s.pool && hipPeekAtLastError() == hipSuccess) c->pool_dirty = false;   // (its last workgroup folds the ticket into the state and zeroes it)
}

// views mode through line entries: pass A leaves a 4-byte entry per line and pass B never reads the input again.  With
// validation the entries carry two flag bits per line (non-ascii byte, byte outside the quality range); only the
// reference's SIMD-width quirk of the quality check (compat_simd_width) depends on WHERE in a line a byte
// sits and keeps the byte-level kernels.
bool views_validating(const bzq_ctx* c) { return c->cfg.check_ascii || c->cfg.check_quality; }
bool views_meta(const bzq_ctx* c) {
    return c->cfg.views_only && !c->views_bytes && !c->views_bytes_once &&
           !(c->cfg.check_quality && c->cfg.compat_simd_width != 0);
}

void launch_views(bzq_ctx* c, dim3 grid, int64_t tb, int64_t te) {
    if (views_meta(c)) {
        const bool growth = c->cfg.buffer_growth_enabled != 0;
        JoinArgs j{c->cur, c->cur_prev_byte, (int64_t)c->cur_n, tb, te, tiles_for(c->cur_n), (const uint32_t*)c->tile_c.p,
                   (const u64*)c->tile_idc.p, (const int64_t*)c->tileP.p, (const uint32_t*)c->entries.p,
                   (const uint32_t*)c->tile_list.p, (int64_t*)c->o().off[0].p, (int64_t*)c->o().off[1].p, (int64_t*)c->o().off[2].p,
                   (int64_t*)c->o().off[3].p, (int64_t*)c->o().rec_end.p, (int64_t*)c->o().id_start.p, (int32_t*)c->o().id_len.p, c->o().rec_cap,
                   c->cur_first_header, growth ? c->cfg.buffer_max_capacity : c->cfg.buffer_capacity, c->d_state,
                   (const uint8_t*)c->tile_vf.p, c->cfg.check_ascii, c->cfg.check_quality};
        const dim3 jg((unsigned)((te - tb + JOIN_TILES - 1) / JOIN_TILES));
        if (views_validating(c)) hipLaunchKernelGGL(k_views_join<true>, jg, dim3(BLOCK), 0, c->stream, j);
        else hipLaunchKernelGGL(k_views_join<false>, jg, dim3(BLOCK), 0, c->stream, j);
        return;
    }
    ViewArgs v{c->cur, (int64_t)c->cur_n, c->cur_prev_byte, tb, te, (const int64_t*)c->tileP.p,
               (int64_t*)c->o().off[0].p, (int64_t*)c->o().off[1].p, (int64_t*)c->o().off[2].p, (int64_t*)c->o().off[3].p,
               (int64_t*)c->o().rec_end.p, (int64_t*)c->o().id_start.p, (int32_t*)c->o().id_len.p, c->o().rec_cap, c->d_state,
               (uint32_t)c->cfg.q_lower, (uint32_t)c->cfg.q_upper, c->force_dense};
    const bool ca = c->cfg.check_ascii != 0, cq = c->cfg.check_quality != 0;
    if (ca && cq) hipLaunchKernelGGL((k_views<true, true>), grid, dim3(BLOCK), 0, c->stream, v);
    else if (ca) hipLaunchKernelGGL((k_views<true, false>), grid, dim3(BLOCK), 0, c->stream, v);
    else if (cq) hipLaunchKernelGGL((k_views<false, true>), grid, dim3(BLOCK), 0, c->stream, v);
    else hipLaunchKernelGGL((k_views<false, false>), grid, dim3(BLOCK), 0, c->stream, v);
}

// Enqueue aggregate -> scan -> emit for every pass of the current chunk.  `emit_only` re-runs just
// the emit kernels (after the per-record arrays were re-sized).
int enqueue_passes(bzq_ctx* c, bool emit_only, bool skip_aggregate_mid) {
    const int64_t nt = tiles_for(c->cur_n);
    const int64_t pt = pass_tiles(c);
    int64_t passes = 0;
    if (!emit_only && skip_aggregate_mid) {
        // shard whose aggregates exist already (bzq_shard_scan): only tile 0 (prev byte now known)
        // and the tiles touched by the appended halo change
        AggArgs a{c->cur, (int64_t)c->cur_n, c->cur_prev_byte, 0, 1, (uint32_t*)c->tile_c.p, (u64*)c->tile_a.p,
                  (u64*)c->tile_idc.p, walk_limit_of(c), (u64*)c->tile_last.p};
        hipLaunchKernelGGL(k_tile_aggregate2, dim3(1), dim3(BLOCK), 0, c->stream, a);
        const int64_t tb = std::max<int64_t>(1, (int64_t)(c->agg_n / TILE));
        if (tb < nt) {
            a.tile_begin = tb; a.tile_end = nt;
            hipLaunchKernelGGL(k_tile_aggregate2, dim3((unsigned)(nt - tb)), dim3(BLOCK), 0, c->stream, a);
        }
    }
    for (int64_t tb = 0; tb < nt; tb += pt, ++passes) {
        const int64_t te = std::min(nt, tb + pt);
        const dim3 grid((unsigned)(te - tb));
        if (!emit_only) {
            if (c->timing_detail) mark_detail(c, passes == 0 ? c->ev[0] : nullptr);   // (the first pass starts where the submit's clock starts)
            if (!skip_aggregate_mid) {
                AggArgs a{c->cur, (int64_t)c->cur_n, c->cur_prev_byte, tb, te, (uint32_t*)c->tile_c.p,
                          (u64*)c->tile_a.p, (u64*)c->tile_idc.p, walk_limit_of(c), (u64*)c->tile_last.p};
                if (views_meta(c)) {
                    LineArgs la{c->cur, (int64_t)c->cur_n, c->cur_prev_byte, tb, te, (uint32_t*)c->tile_c.p, (u64*)c->tile_a.p,
                                (u64*)c->tile_idc.p, (uint32_t*)c->entries.p, (uint32_t*)c->tile_list.p, c->pool_slots, c->d_pool,
                                c->force_dense, (uint8_t*)c->tile_vf.p, (uint32_t)c->cfg.q_lower, (uint32_t)c->cfg.q_upper};
                    // the ticket is zero between chunks: the scan's last workgroup folds and zeroes it -- unless a submit died between
                    // pass A and its scan; then it is zeroed here, on the ctx stream, in front of the next pass A
                    if (c->pool_dirty) HIPCHK(c, hipMemsetAsync(c->d_pool, 0, sizeof(ViewsPool), c->stream));
                    c->pool_dirty = true;
                    const dim3 lg((unsigned)((te - tb + LINES_TPW - 1) / LINES_TPW));   // a workgroup walks LINES_TPW tiles
                    if (views_validating(c)) hipLaunchKernelGGL(k_tile_lines<true>, lg, dim3(BLOCK), 0, c->stream, la);
                    else hipLaunchKernelGGL(k_tile_lines<false>, lg, dim3(BLOCK), 0, c->stream, la);
                } else if (c->cfg.views_only) hipLaunchKernelGGL(k_tile_count, grid, dim3(BLOCK), 0, c->stream, a);
#if BZQ_EXPERIMENTS
                else if (!c->v2) hipLaunchKernelGGL(k_tile_aggregate, grid, dim3(BLOCK), 0, c->stream, a);
#endif
                else if (c->pass_a_h && !c->exact_pass_a && c->exact_sticky == 0) { hipLaunchKernelGGL(k_tile_aggregate_h, grid, dim3(BLOCK), 0, c->stream, a); c->used_h = true; }
                else hipLaunchKernelGGL(k_tile_aggregate2, grid, dim3(BLOCK), 0, c->stream, a);
            }
            if (c->timing_detail) mark_detail(c);
            if (c->init_deferred) {   // pass A is on its way: now the state's initial values, on the side stream
                c->init_deferred = false;
                HIPCHK(c, hipMemcpyAsync(c->d_state, c->h_state, sizeof(ChunkState), hipMemcpyHostToDevice, c->stream_init));
                HIPCHK(c, hipEventRecord(c->ev_init, c->stream_init));
                c->init_in_flight = true;
            }
            if (c->init_in_flight) { (void)hipStreamWaitEvent(c->stream, c->ev_init, 0); c->init_in_flight = false; }
            launch_scan(c, tb, te, (int)passes);
            if (c->fold) launch_batch_bases(c);
            if (c->timing_detail) mark_detail(c);
        } else if (c->fold) {
            // re-run after the per-record arrays were re-sized: the batch table grew with them, so the scan notes the boundary
            // tiles again (same prefixes as before) and the bases are computed again
            ensure_bb(c);
            if (c->o().bb_cap == 0 || ensure(c, c->btile, (size_t)c->o().bb_cap * 8) != 0) { c->err = "batch table after a re-size"; return BZQ_ERR_NOMEM; }
            launch_scan(c, tb, te, (int)passes);
            launch_batch_bases(c);
        }
        if (c->cfg.views_only) {
            launch_views(c, grid, tb, te);
#if BZQ_EXPERIMENTS
        } else if (!c->v2) {
            EmitArgs e = make_emit_args(c);
            e.tile_begin = tb;
            launch_emit(c, grid, e);
#endif
        } else {
            FusedArgs f = make_fused_args(c);
            f.tile_begin = tb; f.tile_end = te;
            if (c->overlap && !emit_only) {
                // emit(k) runs on the second stream behind scan(k); pass A of k+1 proceeds on the main stream
                while ((int64_t)c->ev_pipe.size() <= 2 * passes + 1) { hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming); c->ev_pipe.push_back(ev); }
                if (!c->stream2 && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate (second stream of the sub-chunk passes)"; return BZQ_ERR_HIP; }
                (void)hipEventRecord(c->ev_pipe[2 * passes], c->stream);
                (void)hipStreamWaitEvent(c->stream2, c->ev_pipe[2 * passes], 0);
                hipStream_t keep = c->stream;
                c->stream = c->stream2;
                launch_fused<false>(c, grid, f);
                c->stream = keep;
                (void)hipEventRecord(c->ev_pipe[2 * passes + 1], c->stream2);
            } else {
                launch_fused<false>(c, grid, f);
            }
        }
        if (!emit_only && c->timing_detail) mark_detail(c);
    }
    if (c->overlap && !emit_only && c->v2)
        for (int64_t k = 0; k < passes; ++k) (void)hipStreamWaitEvent(c->stream, c->ev_pipe[2 * k + 1], 0);
    c->n_passes = passes;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("kernel launch: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    return 0;
}

void enqueue_rebase(bzq_ctx* c) {
    const bool growth = c->cfg.buffer_growth_enabled != 0;
    if (views_meta(c)) return;   // the join kernel did the per-record checks and the chunk totals
    if (c->cfg.views_only) {
        ViewCheckArgs va{(const int64_t*)c->o().off[0].p, (const int64_t*)c->o().off[1].p, (const int64_t*)c->o().off[2].p,
                         (const int64_t*)c->o().off[3].p, (const int64_t*)c->o().rec_end.p, c->cur_first_header,
                         growth ? c->cfg.buffer_max_capacity : c->cfg.buffer_capacity, c->o().rec_cap, c->d_state, c->cur,
                         c->cfg.check_quality ? c->cfg.compat_simd_width : 0, (uint32_t)c->cfg.q_upper};
        hipLaunchKernelGGL(k_views_check, dim3((unsigned)(c->num_cu * 8)), dim3(BLOCK), 0, c->stream, va);
        return;
    }
    // batch boundaries for bzq_batch_view (host cache)
    if (!c->fold) ensure_bb(c);
    const int64_t nb_max = c->o().rec_cap / std::max<int64_t>(1, c->cfg.batch_size) + 2;
    int64_t* bb = c->o().bb_cap ? (int64_t*)c->o().bb.p : nullptr;
    if (c->fold) {
        if (!c->finish_done) hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, c->stream, finish_args(c, true), c->d_state);
        c->finish_done = false;
    } else {
    RebaseArgs ra{(const int64_t*)c->o().ends.p, (const int64_t*)c->o().id_ends.p, (const int64_t*)c->o().rec_end.p,
                  (int64_t*)c->o().b_ends.p, (int64_t*)c->o().b_id_ends.p, (int64_t)c->cfg.batch_size, c->cur_first_header,
                  growth ? c->cfg.buffer_max_capacity : c->cfg.buffer_capacity, c->o().rec_cap, c->d_state, c->cur,
                  c->cfg.check_quality ? c->cfg.compat_simd_width : 0, (uint32_t)c->cfg.q_upper, bb, c->o().bb_cap};
    hipLaunchKernelGGL(k_rebase, dim3((unsigned)(c->num_cu * 8)), dim3(BLOCK), 0, c->stream, ra);
    }
    // mirror the batch-boundary table into pinned memory on the stream: it arrives with the chunk state, no extra synchronisation
    OutSet& o = c->o();
    o.h_bb_batches = 0;
    if (bb) {
        ensure_h_bb(c);
        // (at most what the chunk can hold: n / min_record_bytes records; the count is only known after the kernels)
        if (c->published) { if (o.h_bb) o.h_bb_batches = std::min(nb_max, o.h_bb_cap); }   // k_tail wrote the batches that exist
        else if (o.h_bb && hipMemcpyAsync(o.h_bb, bb, (size_t)nb_max * 16, hipMemcpyDeviceToHost, c->stream) == hipSuccess) o.h_bb_batches = nb_max;
    }
}

// the side stream of the state's initial values (option state_init_in_kernel = 0, and chunks parsed in several passes): created on
// first use, IN THE PARSER'S PRIORITY CLASS -- a stream of the default class shares the hardware queues of that class with a caller's
// streams and the scan, which waits for this copy, can end up behind somebody's long kernel (round 5: BGZF ingest 30 instead of 44 GB/s)
static bool ensure_init_stream(bzq_ctx* c) {
    if (c->stream_init && c->ev_init) return true;
    if (c->init_stream_failed) return false;
    if (!c->stream_init && (c->stream_has_prio ? hipStreamCreateWithPriority(&c->stream_init, hipStreamNonBlocking, c->stream_prio)
                                               : hipStreamCreateWithFlags(&c->stream_init, hipStreamNonBlocking)) != hipSuccess) { c->stream_init = nullptr; (void)hipGetLastError(); }
    if (!c->ev_init && hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming) != hipSuccess) { c->ev_init = nullptr; (void)hipGetLastError(); }
    c->init_stream_failed = !(c->stream_init && c->ev_init);
    return !c->init_stream_failed;
}

int submit_common(bzq_ctx* c, const uint8_t* d_data, uint64_t n, uint64_t stream_pos, int is_eof,
                  int64_t P0, int64_t S0, int64_t Q0, int64_t I0, uint32_t prev_byte, int64_t first_header,
                  const int64_t* first_nl = nullptr, int head_lines = 0, bool reuse_aggregates = false) {
    int rc;
    if ((rc = begin_submit(c))) return rc;
    if (c->consumer_guard && c->consumer_stream && c->consumer_stream != c->stream) {
        // the set this submit writes held the chunk before the previous one: its consumers were on the consumer stream when the
        // previous submit recorded its event
        for (hipEvent_t& e : c->ev_guard) if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (c->guard_serial > 0) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_guard[(c->guard_serial - 1) & 1], 0));
        HIPCHK(c, hipEventRecord(c->ev_guard[c->guard_serial & 1], c->consumer_stream));
        c->guard_serial += 1;
    }
    if ((rc = ensure_tile_arenas(c, n)) || (rc = ensure_col_arenas(c, n))) return rc;
    int64_t want = (int64_t)(n / (uint64_t)std::max(4, c->cfg.min_record_bytes)) + 1024;
    if ((rc = ensure_record_arenas(c, want))) return rc;
    c->views_bytes_once = false; c->tail_pending = false; c->cum_valid = false;
    c->o().cum_valid = false; c->o().parsed = false; c->o().fold = false;
    if (!reuse_aggregates) c->used_h = false;   // (a shard's aggregates come from bzq_shard_scan, which says how it made them)
    c->cur = d_data; c->cur_n = n; c->cur_stream_pos = stream_pos; c->cur_is_eof = is_eof;
    c->cur_prev_byte = prev_byte; c->cur_first_header = first_header;
    c->ev_detail.clear(); c->ev_detail_used = 0;
    ChunkState* h = c->h_state;
    memset(h, 0, sizeof(*h));
    h->P0 = P0; h->S0 = S0; h->Q0 = Q0; h->I0 = I0;
    h->P = P0; h->S = S0; h->Q = Q0; h->I = I0;
    h->last_nl_tile = -1; h->tail_start = 0;
    h->err_struct = ~0ull; h->err_valid = ~0ull; h->err_buf = ~0ull;
    for (int i = 0; i < 4; ++i) h->first_nl[i] = first_nl ? first_nl[i] : -1;
    // Two-pass path from the chunk's first byte: pass A never looks at the state, so its initial values travel beside it and the
    // scan waits for them (enqueue_passes) -- otherwise the copy sits in front of the kernels.
    c->init_in_flight = false; c->init_deferred = false; c->published = false;
    // Two-pass path from the chunk's first byte, one pass over the chunk: NO copy at all -- the first workgroup of k_scan_reduce
    // writes the state's initial values (pass A does not look at the state; everything that does comes behind the scan).  Round 5 tried
    // this and dropped it over a wrong answer that turned out to be bzq_create's; round 6 made it the default: one stream
    // and one event less per ctx (a stream costs 9-20 ms to create: scripts/probes/create_probe.hip), one copy packet less per submit.
    // Option state_init_in_kernel = 0: the copy on a side stream (created on first use), as rounds 5-6 shipped it.
    bool plain = c->lean && n > 0 && head_lines == 0 && !first_nl && !reuse_aggregates &&
                 (c->single_pass == 0 || c->cfg.views_only) && !c->use_stream && !c->overlap;
    const bool by_kernel = plain && c->init_in_kernel && pass_tiles(c) >= tiles_for(n);
    if (plain && !by_kernel && !ensure_init_stream(c)) plain = false;   // (no side stream: the copy stays on the ctx stream)
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    c->init_by_kernel_now = false;
    if (by_kernel) {
        c->h_init[0] = P0; c->h_init[1] = S0; c->h_init[2] = Q0; c->h_init[3] = I0;
        if (c->init_in_kernel == 4) hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, c->stream, c->d_state, P0, S0, Q0, I0);
        else c->init_by_kernel_now = true;
    } else if (plain) {
        c->init_deferred = true;   // (enqueue_passes, behind pass A's launch)
    } else {
        HIPCHK(c, hipMemcpyAsync(c->d_state, h, sizeof(ChunkState), hipMemcpyHostToDevice, c->stream));
    }
    if (head_lines > 0)
        hipLaunchKernelGGL(k_head, dim3(1), dim3(64), 0, c->stream, d_data, (int64_t)n, prev_byte, head_lines, c->d_state);
    if (n > 0) {
        c->ran_single_pass = c->single_pass != 0 && !c->cfg.views_only;   // views mode has only the two-pass kernels
        // one read of the input (k_stream) unless the caller asked for sub-chunk passes, or this is a shard (its pass A
        // already ran in bzq_shard_scan and its first lines belong to the previous rank)
        c->ran_stream = !c->ran_single_pass && c->use_stream && !c->cfg.views_only && c->cfg.pass_bytes == 0 && head_lines == 0 &&
                        !reuse_aggregates && !first_nl;
        decide_fold(c);
        if (c->fold && plain) { ensure_h_bb(c); c->published = c->o().h_bb != nullptr; }
        if (c->ran_single_pass) { if ((rc = enqueue_single_launch(c))) return rc; }
        else if (c->ran_stream) { if ((rc = enqueue_stream(c))) return rc; }
        else {
            if ((rc = enqueue_passes(c, false, reuse_aggregates))) return rc;
            if (c->exact_sticky > 0 && !c->cfg.views_only) c->exact_sticky -= 1;
        }
        // (views mode through line entries: the join's last workgroup looks at the tail itself)
        if (!(views_meta(c) && !c->ran_single_pass && !c->ran_stream))
            hipLaunchKernelGGL(k_tail, dim3(1), dim3(BLOCK), 0, c->stream, c->cur, (int64_t)n, c->d_state, finish_args(c, true));
        c->finish_done = c->fold;
    } else c->fold = false;
    enqueue_rebase(c);
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));   // ev[0] .. ev[3]: the submit's kernels (two events per submit; option timing_detail adds three)
    if (!c->published) {
        // views mode on the plain path: a one-workgroup kernel writes the state into the pinned copy (no copy packet); everything else copies
        if (plain && n > 0 && c->cfg.views_only && c->lean) hipLaunchKernelGGL(k_publish_state, dim3(1), dim3(128), 0, c->stream, (const ChunkState*)c->d_state, c->h_state);
        else HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(ChunkState), hipMemcpyDeviceToHost, c->stream));
    }
    c->pending = true; c->have_result = false;
    return follow_consume(c);   // (host work: it overlaps the parse that was just enqueued)
}

// Replays the reference's BufferedReader over the delivered records to learn the window state at
// the moment the parser reaches the trailing non-record bytes, then classifies them exactly as
// _next_ref_complete does (parser.mojo:451-522).  Cold path: only for a stream that ends in junk.
// In two halves so that the multi-GPU protocol can walk the window through the ranks' records in rank order
// (bzq_shard_stitch): window_walk advances (s, head) over a run of record ends, window_classify judges the tail.
void window_start(Window& s, const bzq_config& cfg, int64_t N) {
    s = Window();
    s.N = N; s.cap = cfg.buffer_capacity; s.w = 0; s.end = 0; s.eof = false;
    s.fill(); // BufferedReader.__init__, buffered.mojo:149
}
void window_walk(Window& s, int64_t& head, const int64_t* rec_end, size_t n, int64_t offset, const bzq_config& cfg) {
    const bool growth = cfg.buffer_growth_enabled != 0;
    const int64_t maxcap = cfg.buffer_max_capacity;
    for (size_t r = 0; r < n; ++r) {
        const int64_t E = rec_end[r] + offset;
        if (head == s.end) { s.w = head; s.fill(); }
        while (!(E < s.end)) {
            if (head == s.w) {
                if (!growth || s.cap >= maxcap) break; // cannot happen for delivered records (err_buf caught it)
                s.cap = std::min(maxcap, s.cap + std::min(s.cap, maxcap - s.cap));
            } else {
                s.w = head;
            }
            s.fill();
        }
        head = E + 1;
    }
}
int window_classify(Window& s, int64_t head, const bzq_config& cfg, int tail_phase, bool tail_nonblank, bool* accept_last,
                    int* phase_out, int64_t* cap_out) {
    const bool growth = cfg.buffer_growth_enabled != 0;
    const int64_t maxcap = cfg.buffer_max_capacity;
    *accept_last = false;
    if (head == s.end) { s.w = head; s.fill(); }
    if (head == s.end && s.eof) return BZQ_EOF;
    for (;;) {
        const int64_t avail = s.end - head;
        if (avail < s.cap && s.eof) {
            if (tail_phase == 3) {
                if (!tail_nonblank) return BZQ_OTHER; // `raise Error()` with an empty message, parser.mojo:350-351
                *accept_last = true;
                return BZQ_OK;
            }
            *phase_out = tail_phase;
            return BZQ_UNEXPECTED_EOF;
        }
        if (head == s.w) {
            if (!growth) { *cap_out = s.cap; return BZQ_BUFFER_EXCEEDED; }
            if (s.cap >= maxcap) { *cap_out = maxcap; return BZQ_BUFFER_AT_MAX; }
            s.cap = std::min(maxcap, s.cap + std::min(s.cap, maxcap - s.cap));
        } else {
            s.w = head;
        }
        const int64_t filled = s.fill();
        if (filled == 0 && s.end - head == 0) return BZQ_EOF;
    }
}
int classify_tail(bzq_ctx* c, const std::vector<int64_t>& rec_end, int64_t N, int64_t first_header, int64_t consumed,
                  int tail_phase, bool tail_nonblank, bool* accept_last, int* phase_out, int64_t* cap_out) {
    Window s;
    window_start(s, c->cfg, N);
    int64_t head = first_header;
    window_walk(s, head, rec_end.data(), rec_end.size(), 0, c->cfg);
    return window_classify(s, consumed, c->cfg, tail_phase, tail_nonblank, accept_last, phase_out, cap_out);
}

constexpr int64_t FOLLOW_UNBOUNDED = 1ll << 60;
// The caller has told (records_before of the chunk being submitted) how many records of the previous chunk it took: walk the
// window over exactly those.  Comparisons of the walk only involve records that exist, so the stream's still unknown length
// does not enter (Window::N = unbounded; window_classify gets the true length, and `end` clipped to it, at the stream's end).
int follow_consume(bzq_ctx* c) {
    if (c->records_before < 0 || c->shard_mode) { c->stage_valid = false; c->follow_on = false; return 0; }
    if (c->records_before == 0) { c->stage_valid = false; c->follow_on = false; return 0; }   // a new stream starts with this chunk
    if (!c->stage_valid) return 0;   // (nothing staged: the follower stays where it is, or off)
    const int64_t taken = std::max<int64_t>(0, std::min<int64_t>(c->stage_n, c->records_before - c->stage_rb));
    HIPCHK(c, hipEventSynchronize(c->stage_ev));
    if (c->stage_rb == 0 || !c->follow_on) {   // the stream's first chunk: BufferedReader.__init__
        if (c->stage_rb != 0) { c->stage_valid = false; return 0; }   // (a stream picked up in the middle: no window to follow)
        window_start(c->follow, c->cfg, FOLLOW_UNBOUNDED);
        c->follow_head = c->stage_head;
        c->follow_records = 0;
        c->follow_on = true;
    }
    if (c->follow_records != c->stage_rb) { c->follow_on = false; c->stage_valid = false; return 0; }   // records_before jumped: not one stream
    window_walk(c->follow, c->follow_head, c->stage_pin, (size_t)taken, c->stage_pos, c->cfg);
    c->follow_records += taken;
    c->stage_valid = false;
    return 0;
}
// Behind a parsed chunk of a multi-chunk stream: its record ends to the host, asynchronously
int follow_stage(bzq_ctx* c, int64_t n_complete) {
    c->stage_valid = false;
    if (c->records_before < 0 || c->shard_mode) return 0;
    if ((size_t)n_complete > c->stage_cap) {
        if (c->stage_pin) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipHostFree(c->stage_pin)); c->stage_pin = nullptr; c->stage_cap = 0; }
        const size_t want = (size_t)n_complete + (size_t)n_complete / 4 + 1024;
        if (hipHostMalloc((void**)&c->stage_pin, want * 8, hipHostMallocDefault) != hipSuccess) { c->err = "pinned staging for the record ends"; return BZQ_ERR_NOMEM; }
        c->stage_cap = want;
    }
    if (!c->stage_ev) HIPCHK(c, hipEventCreateWithFlags(&c->stage_ev, hipEventDisableTiming));
    if (n_complete > 0) HIPCHK(c, hipMemcpyAsync(c->stage_pin, c->o().rec_end.p, (size_t)n_complete * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->stage_ev, c->stream));
    c->stage_n = n_complete; c->stage_rb = c->records_before;
    c->stage_pos = (int64_t)c->cur_stream_pos;
    c->stage_head = (int64_t)c->cur_stream_pos + c->cur_first_header;
    c->stage_valid = true;
    return 0;
}

void sb_put(std::string& s, const char* label, long long v) {
    s += label;
    s += std::to_string(v);
}

} // namespace

// ================================================================================== C ABI

#include "bzq_consumers.hpp"
#include "bzq_inflate.hpp"
#if BZQ_EXPERIMENTS
#include "bzq_inflate_ms.hpp"   // experiments/csrc: eight BGZF blocks per wave (correct, 2.7x slower: profiles/r4_inflate_ms.md)
#endif
#include "bzq_gzip.hpp"
#include "bzq_ingest.hpp"

extern "C" {

int32_t bzq_abi_version(void) { return BZQ_ABI_VERSION; }

void bzq_config_default(bzq_config* cfg) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->buffer_capacity = 256 * 1024;       // CONSTS.mojo:26
    cfg->buffer_max_capacity = 1ll << 30;    // CONSTS.mojo:28
    cfg->q_lower = 33; cfg->q_upper = 126; cfg->q_offset = 33; // generic_schema
    cfg->batch_size = 4096;                  // CONSTS.mojo:31
    cfg->max_chunk_bytes = 256ll << 20;
    cfg->pass_bytes = 0;
    cfg->min_record_bytes = 32;
}

int32_t bzq_schema_from_name(const char* name, uint8_t* lower, uint8_t* upper, uint8_t* offset) {
    // utils.mojo:612-637 over quality_schema.mojo:26-31
    struct Row { const char* n; uint8_t lo, up, off; };
    static const Row rows[] = {{"sanger", 33, 126, 33},       {"solexa", 59, 126, 64},
                               {"illumina_1.3", 64, 126, 64}, {"illumina_1.5", 66, 126, 64},
                               {"illumina_1.8", 33, 126, 33}, {"generic", 33, 126, 33}};
    for (const Row& r : rows)
        if (name && strcmp(name, r.n) == 0) { *lower = r.lo; *upper = r.up; *offset = r.off; return 1; }
    *lower = 33; *upper = 126; *offset = 33;
    return 0;
}

const char* bzq_message_for_code(int32_t code) { return message_for_code(code); }

int32_t bzq_host_simd_width(void) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512bw")) return 64;
    if (__builtin_cpu_supports("avx2")) return 32;
#endif
    return 16;
}

int32_t bzq_create(int32_t device, const bzq_config* cfg, bzq_ctx** out) {
    if (!cfg || !out) return BZQ_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (libblazeseq_hip has no CPU fallback)";
        return BZQ_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_error = "device ordinal out of range"; return BZQ_ERR_ARG; }
    if (cfg->batch_size <= 0 || cfg->buffer_capacity <= 0 || cfg->q_upper > 127 || cfg->q_lower > 128 ||
        cfg->q_lower > cfg->q_upper ||
        !(cfg->compat_simd_width == 0 || cfg->compat_simd_width == 16 || cfg->compat_simd_width == 32 ||
          cfg->compat_simd_width == 64)) {
        g_create_error = "invalid bzq_config";
        return BZQ_ERR_ARG;
    }
    bzq_ctx* c = new bzq_ctx();
    c->device = device;
    c->cfg = *cfg;
    if (c->cfg.min_record_bytes <= 0) c->cfg.min_record_bytes = 32;
    if (c->cfg.pass_bytes > 0) c->cfg.pass_bytes = std::max<int64_t>(TILE, (c->cfg.pass_bytes / TILE) * TILE);
#define CRT(call)                                                                     \
    do {                                                                              \
        hipError_t e2 = (call);                                                       \
        if (e2 != hipSuccess) {                                                       \
            g_create_error = std::string(#call) + ": " + hipGetErrorString(e2);       \
            delete c;                                                                 \
            return BZQ_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)
    CRT(hipSetDevice(device));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
            g_create_error = std::string("bzq_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only (no other code path exists)";
            delete c;
            return BZQ_ERR_NO_DEVICE;
        }
        if (prop.multiProcessorCount > 0) c->num_cu = prop.multiProcessorCount;
    }
    {   // the parser's stream: highest priority.  Its kernels are short and sit on the critical path of a file-backed stream, beside
        // 10 ms inflate kernels that fill the device; and the runtime keeps a pool of hardware queues per priority
        // (GPU_MAX_HW_QUEUES = 4, further streams SHARE them): a stream of its own class does not land in a queue behind a helper
        // stream's long kernel (rocprofv3 timeline of the BGZF ingest, round 4: the parse of chunk k waited for the inflate of k + 1)
        int lo = 0, hi = 0;
        const char* e = getenv("BZQ_STREAM_PRIORITY");   // (measurements: 0 = a stream of the default class)
        if ((!e || e[0] != '0') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) { CRT(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi)); c->stream_prio = hi; c->stream_has_prio = true; }
        else CRT(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    CRT(hipMalloc((void**)&c->d_state, sizeof(ChunkState)));
    CRT(hipMalloc((void**)&c->d_pool, sizeof(ViewsPool)));
    // diagnostics of the create-time race hunt: BZQ_POOL_POISON = fill the ticket with garbage first (so that a zeroing
    // that comes late is SEEN); BZQ_POOL_ZERO = 0: hipMemset on the NULL stream as rounds 4-5 did, 1 (default): on the ctx stream
    {
        const char* pz = getenv("BZQ_POOL_ZERO");
        if (getenv("BZQ_POOL_POISON")) CRT(hipMemset(c->d_pool, 0x7F, sizeof(ViewsPool)));
        if (pz && pz[0] == '0') CRT(hipMemset(c->d_pool, 0, sizeof(ViewsPool)));
        else CRT(hipMemsetAsync(c->d_pool, 0, sizeof(ViewsPool), c->stream));
    }
    CRT(hipHostMalloc((void**)&c->h_state, sizeof(ChunkState), hipHostMallocDefault));
    for (auto& ev : c->ev) CRT(hipEventCreate(&ev));
    { const char* e = getenv("BZQ_LEAN_SUBMIT"); if (e && e[0] == '0') c->lean = 0; }   // (bisecting aid: option lean_submit for a whole process)
    { const char* e = getenv("BZQ_STATE_INIT"); if (e && e[0] >= '0' && e[0] <= '4') c->init_in_kernel = e[0] - '0'; }   // (the same for option state_init_in_kernel)
    // (the side stream of the state's initial values is created on first use: ensure_init_stream)
#undef CRT
    *out = c;
    return 0;
}

int32_t bzq_comm_destroy(bzq_ctx* c);

void bzq_destroy(bzq_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)bzq_comm_destroy(c);
    std::vector<DevBuf*> bufs = {&c->in, &c->tile_c, &c->tile_a, &c->tile_idc, &c->tileP, &c->tileS, &c->tileQ, &c->tileI, &c->grp, &c->desc,
                                 &c->consumer_scratch, &c->qpos_scratch, &c->gen_prefix, &c->entries, &c->tile_list, &c->tile_vf, &c->inflate_tab, &c->tile_last, &c->tileB, &c->btile, &c->shard_buf};
    for (OutSet& o : c->out) {
        for (DevBuf* b : {&o.seq, &o.qual, &o.id, &o.ends, &o.id_ends, &o.rec_end, &o.b_ends, &o.b_id_ends, &o.off[0], &o.off[1],
                          &o.off[2], &o.off[3], &o.id_start, &o.id_len, &o.bb})
            bufs.push_back(b);
        for (DevBuf& b : o.view_blocks) bufs.push_back(&b);
        if (o.h_bb) (void)hipHostFree(o.h_bb);
    }
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    for (void* q : c->shard_pin) bzq::cache::pinned_pool().put(q);
    for (hipStream_t q : c->shard_streams) if (q) (void)hipStreamDestroy(q);
    for (hipEvent_t q : c->shard_events) if (q) (void)hipEventDestroy(q);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_pool) (void)hipFree(c->d_pool);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->stage_pin) (void)hipHostFree(c->stage_pin);
    if (c->stage_ev) (void)hipEventDestroy(c->stage_ev);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t e : c->ev_detail_pool) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream_init) (void)hipStreamDestroy(c->stream_init);
    if (c->ev_init) (void)hipEventDestroy(c->ev_init);
    for (hipEvent_t e : c->ev_guard) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_pipe) (void)hipEventDestroy(e);
    delete c;
}

const char* bzq_last_error(const bzq_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int32_t bzq_set_stream(bzq_ctx* c, void* hip_stream) {
    if (!c) return BZQ_ERR_ARG;
    if (c->pending) { c->err = "bzq_set_stream while a chunk is in flight"; return BZQ_ERR_ARG; }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return 0;
}

int32_t bzq_set_consumer_stream(bzq_ctx* c, void* hip_stream) {
    if (!c) return BZQ_ERR_ARG;
    c->consumer_stream = (hipStream_t)hip_stream;
    return 0;
}

int32_t bzq_get_config(const bzq_ctx* c, bzq_config* out) {
    if (!c || !out) return BZQ_ERR_ARG;
    *out = c->cfg;
    return 0;
}

static void gpu_numa_cpus(int device, int* node_out, std::vector<int>& cpus);
int32_t bzq_set_option(bzq_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return BZQ_ERR_ARG;
    if (!strcmp(key, "force_dense")) c->force_dense = (int)value;
    else if (!strcmp(key, "timing_detail")) c->timing_detail = (int)value;
    else if (!strcmp(key, "single_pass") || !strcmp(key, "kernels_v2")) {
        // the single-launch variants and the first-generation kernels are cross-checks, compiled only with EXPERIMENTS=1
        const bool dflt = !strcmp(key, "single_pass") ? value == 0 : value != 0;
        if (!BZQ_EXPERIMENTS && !dflt) { c->err = std::string("option ") + key + " needs a library built with EXPERIMENTS=1"; return BZQ_ERR_ARG; }
        if (!strcmp(key, "single_pass")) c->single_pass = (int)value; else c->v2 = (int)value;
    }
    else if (!strcmp(key, "stream")) {
        if (!BZQ_EXPERIMENTS && value) { c->err = "option stream needs a library built with EXPERIMENTS=1"; return BZQ_ERR_ARG; }
        c->use_stream = value != 0;
    }
    else if (!strcmp(key, "stream_fallbacks")) return (int32_t)std::min<int64_t>(c->stream_fallbacks, 0x7FFFFFFF);   // query: chunks repeated on the two-pass kernels
    else if (!strcmp(key, "last_folded")) return c->fold ? 1 : 0;                                                     // query: the last chunk went without k_rebase
    else if (!strcmp(key, "dump_state")) {   // query (diagnostic): the last result's state snapshots to stderr
        for (int i = 0; i < 3; ++i) if (c->dbg[i][11])
            fprintf(stderr, "bzq state[%s]: err_struct %lld overflow %lld P %lld n_complete %lld err_valid %lld err_buf %lld P0 %lld last_nl_tile %lld fallback %lld tail_start %lld last_record_end %lld\n",
                    i == 0 ? "first" : i == 1 ? "after views fallback" : "after re-size", (long long)c->dbg[i][0], (long long)c->dbg[i][1], (long long)c->dbg[i][2], (long long)c->dbg[i][3],
                    (long long)c->dbg[i][4], (long long)c->dbg[i][5], (long long)c->dbg[i][6], (long long)c->dbg[i][7], (long long)c->dbg[i][8], (long long)c->dbg[i][9], (long long)c->dbg[i][10]);
        return 0;
    }
    else if (!strcmp(key, "dense_tiles")) return (int32_t)std::min<int64_t>(c->last_dense_tiles, 0x7FFFFFFF);          // query
    else if (!strcmp(key, "n_submits")) return (int32_t)std::min<int64_t>(c->n_submits, 0x7FFFFFFF);                  // query: chunks submitted so far
    else if (!strcmp(key, "pass_a_h")) c->pass_a_h = value != 0;
    else if (!strcmp(key, "fold_rebase")) c->fold_opt = value != 0;
    else if (!strcmp(key, "lean_submit")) c->lean = value != 0;
    else if (!strcmp(key, "consumer_guard")) c->consumer_guard = value != 0;
    else if (!strcmp(key, "state_init_in_kernel")) c->init_in_kernel = (int)value;
    else if (!strcmp(key, "ranks_seen")) return c->ranks_seen;   // query
    else if (!strcmp(key, "device")) return c->device;           // query
    else if (!strcmp(key, "numa_node") || !strcmp(key, "numa_cpus")) {   // query: the GPU's NUMA node / how many CPUs the reader threads are bound to (0 = not bound)
        int node = -1; std::vector<int> cpus;
        if (c->ingest_numa) gpu_numa_cpus(c->device, &node, cpus);
        return key[5] == 'n' ? (node >= 0 ? node : 255) : (int32_t)cpus.size();   // (255: node unknown)
    }
    else if (!strcmp(key, "cumulative_ends")) c->fold_opt = value == 0;   // 1: bzq_chunk.d_ends / d_id_ends filled with every chunk (the k_rebase path, as before ABI 2)
    else if (!strcmp(key, "pass_a_sticky")) { c->sticky_opt = value != 0; if (!value) { c->exact_sticky = 0; c->exact_sticky_len = 0; } }
    else if (!strcmp(key, "ingest_direct")) c->ingest_direct = value != 0;
    else if (!strcmp(key, "ingest_numa")) c->ingest_numa = value != 0;
    else if (!strcmp(key, "ingest_gpu_inflate")) c->ingest_gpu_inflate = value != 0;
#if BZQ_EXPERIMENTS
    else if (!strcmp(key, "inflate_ms")) c->inflate_ms = value != 0;
#endif
    else if (!strcmp(key, "double_buffer")) {
        if (c->pending) { c->err = "double_buffer cannot change while a chunk is in flight"; return BZQ_ERR_ARG; }
        c->double_buffer = value != 0;
    }
    else if (!strcmp(key, "experiments")) return BZQ_EXPERIMENTS ? 0 : BZQ_ERR_ARG;   // query: is this an EXPERIMENTS build?
    else if (!strcmp(key, "ablate")) c->ablate = (int)value;
    else if (!strcmp(key, "views_bytes")) c->views_bytes = (int)value;
    else if (!strcmp(key, "overlap")) c->overlap = (int)value;
    else if (!strcmp(key, "records_before")) c->records_before = value;
    else if (!strcmp(key, "comm_timeout_ms")) { if (value <= 0) { c->err = "comm_timeout_ms must be positive"; return BZQ_ERR_ARG; } c->comm_timeout_ms = value; }
    else if (!strcmp(key, "pin_cache_bytes") || !strcmp(key, "dev_cache_bytes")) {
        // process-wide (bzq_bufcache.hpp): how many bytes of pinned / device buffers closed streams leave behind for the next open; 0 drops
        // everything held now and turns the cache off
        if (value < 0) { c->err = std::string(key) + " must not be negative"; return BZQ_ERR_ARG; }
        HIPCHK(c, hipSetDevice(c->device));
        (key[0] == 'p' ? bzq::cache::pinned_pool() : bzq::cache::device_pool()).trim((uint64_t)value, false);
    }
    else if (!strcmp(key, "buf_cache_hits")) { const uint64_t h = bzq::cache::pinned_pool().hits_now() + bzq::cache::device_pool().hits_now(); return (int32_t)std::min<uint64_t>(h, 0x7FFFFFFF); }   // query
    else if (!strcmp(key, "buf_cache_held_mb")) return (int32_t)((bzq::cache::pinned_pool().held_now() + bzq::cache::device_pool().held_now()) >> 20);   // query
    else if (!strcmp(key, "pass_bytes")) c->cfg.pass_bytes = value > 0 ? std::max<int64_t>(TILE, (value / TILE) * TILE) : 0;
    else { c->err = std::string("unknown option ") + key; return BZQ_ERR_ARG; }
    return 0;
}

int32_t bzq_pinned_alloc(size_t bytes, void** out) {
    if (!out) return BZQ_ERR_ARG;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? 0 : BZQ_ERR_NOMEM;
}
int32_t bzq_pinned_free(void* p) { return hipHostFree(p) == hipSuccess ? 0 : BZQ_ERR_HIP; }

// ---- BGZF inflate on the device (bzq_inflate.hpp) -------------------------------------------------------------------------

int32_t bzq_bgzf_scan(const uint8_t* comp, uint64_t n, uint64_t max_out, bzq_bgzf_block* blocks, int64_t cap, int64_t* n_blocks,
                      uint64_t* consumed, uint64_t* out_bytes) {
    if (!comp && n) return BZQ_ERR_ARG;
    if (!n_blocks || !consumed || !out_bytes || (cap > 0 && !blocks)) return BZQ_ERR_ARG;
    int64_t k = 0;
    uint64_t off = 0, usum = 0;
    while (off + 28 <= n && k < cap) {
        const uint32_t bs = bzq::bgzf_block_size(comp + off);
        if (!bs || bs < 26) { *n_blocks = k; *consumed = off; *out_bytes = usum; return BZQ_ERR_IO; }
        if (off + bs > n) break;   // the block is not whole yet
        const uint8_t* t = comp + off + bs - 4;
        const uint32_t us = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (us > 65536) { *n_blocks = k; *consumed = off; *out_bytes = usum; return BZQ_ERR_IO; }
        if (usum + us > max_out) break;
        const uint32_t crc = (uint32_t)t[-4] | ((uint32_t)t[-3] << 8) | ((uint32_t)t[-2] << 16) | ((uint32_t)t[-1] << 24);
        blocks[k++] = bzq_bgzf_block{off, bs, us, crc, 0u, usum};
        usum += us; off += bs;
    }
    *n_blocks = k; *consumed = off; *out_bytes = usum;
    return 0;
}

int32_t bzq_bgzf_inflate(bzq_ctx* c, const uint8_t* d_comp, uint64_t comp_bytes, const bzq_bgzf_block* blocks, int64_t n_blocks,
                         uint8_t* d_out, uint64_t out_capacity) {
    if (!c || n_blocks < 0 || (n_blocks && (!d_comp || !blocks || !d_out))) return BZQ_ERR_ARG;
    if (n_blocks == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<bzq::inf::DevBlock> hb((size_t)n_blocks);
    for (int64_t i = 0; i < n_blocks; ++i) {
        const bzq_bgzf_block& b = blocks[i];
        // (subtraction form: an offset near 2^64 must not wrap past the check)
        if (b.comp_size < 26 || b.comp_size > comp_bytes || b.comp_offset > comp_bytes - b.comp_size || b.out_size > 65536 || b.out_size > out_capacity ||
            b.out_offset > out_capacity - b.out_size) {
            c->err = "bzq_bgzf_inflate: block " + std::to_string(i) + " lies outside the buffers";
            return BZQ_ERR_ARG;
        }
        // one wave per block writes its range without looking at the others: the ranges must not overlap
        if (i > 0 && b.out_offset < blocks[i - 1].out_offset + blocks[i - 1].out_size) {
            c->err = "bzq_bgzf_inflate: the output range of block " + std::to_string(i) + " overlaps the block before it (out_offset must not decrease)";
            return BZQ_ERR_ARG;
        }
        hb[(size_t)i] = bzq::inf::DevBlock{b.comp_offset + 18, b.out_offset, b.comp_size - 26, b.out_size, b.crc32, 0u};
    }
    int rc;
    const size_t tbytes = (size_t)n_blocks * sizeof(bzq::inf::DevBlock);
    if ((rc = ensure(c, c->inflate_tab, tbytes + 16))) return rc;
    unsigned long long* d_bad = (unsigned long long*)((uint8_t*)c->inflate_tab.p + ((tbytes + 7) & ~(size_t)7));
    const unsigned long long none = ~0ull;
    HIPCHK(c, hipMemcpyAsync(c->inflate_tab.p, hb.data(), tbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_bad, &none, 8, hipMemcpyHostToDevice, c->stream));
#if BZQ_EXPERIMENTS
    if (c->inflate_ms) {   // (all of that kernel's tables live in LDS: no scratch)
        unsigned long long* d_stats = nullptr;
        if (getenv("BZQ_MS_STATS")) {   // debug: iterations of the symbol loop [0] and services by kind [1..4], printed per call
            if ((rc = ensure(c, c->consumer_scratch, 64))) return rc;
            d_stats = (unsigned long long*)c->consumer_scratch.p;
            HIPCHK(c, hipMemsetAsync(d_stats, 0, 64, c->stream));
        }
        bzq::inf::launch_bgzf_inflate_ms(bzq::inf::ArgsMs{d_comp, comp_bytes, (const bzq::inf::DevBlock*)c->inflate_tab.p, n_blocks, d_out, d_bad, nullptr, d_stats}, c->stream);
        if (d_stats) {
            unsigned long long hs[8];
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipMemcpy(hs, d_stats, 64, hipMemcpyDeviceToHost));
            fprintf(stderr, "bzq inflate_ms: %lld blocks, loop iterations %llu, services: header %llu eob %llu; wave-ms total: service %.1f loop %.1f crc %.1f\n", (long long)n_blocks, hs[0], hs[1], hs[2], hs[5] * 1e-5, hs[6] * 1e-5, hs[7] * 1e-5);
        }
    } else
#endif
    {
        bzq::inf::Args a{d_comp, comp_bytes, (const bzq::inf::DevBlock*)c->inflate_tab.p, n_blocks, d_out, d_bad};
        hipLaunchKernelGGL(bzq::inf::k_bgzf_inflate, dim3((unsigned)((n_blocks + bzq::inf::WAVES - 1) / bzq::inf::WAVES)), dim3(BLOCK), 0, c->stream, a);
    }
    unsigned long long bad = none;
    HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (bad != none) {
        c->err = "BGZF block " + std::to_string(bad) + " failed to inflate (corrupt or truncated file)";
        return BZQ_ERR_IO;
    }
    return 0;
}

// ---- any gzip stream on the device (bzq_gzip.hpp) ---------------------------------------------------------------------------------
int32_t bzq_gzip_open(bzq_ctx* c, bzq_gzip** out) {
    if (!c || !out) return BZQ_ERR_ARG;
    return bzq::gz::gz_open(c->device, out, c->err);
}
int32_t bzq_gzip_set_option(bzq_gzip* h, const char* key, int64_t value) {
    if (!h || !key) return BZQ_ERR_ARG;
    if (!strcmp(key, "chunk_bytes")) {
        if (value < 4096 || value > (1 << 20)) { h->err = "chunk_bytes must lie in [4096, 1 MiB]"; return BZQ_ERR_ARG; }
        h->chunk_bytes = (int32_t)value;
        return 0;
    }
    if (!strcmp(key, "early_find")) { h->early_find = value != 0; return 0; }   // 0: a staged piece's finder runs once the piece in front has been decoded (round 3)
    if (!strcmp(key, "defer_verify")) { h->defer_verify = value != 0; return 0; }   // only for a caller whose consumers are on the handle's stream (bzq_gzip.hpp)
    if (!strcmp(key, "deferred_calls")) return (int32_t)std::min<uint64_t>(h->deferred_calls, 0x7FFFFFFF);   // query
    if (!strcmp(key, "chain_l2")) { h->chain_l2 = value != 0; return 0; }   // the chain kernels beside the next piece's decoders (bzq_gzip.hpp: k_gz_chainl_*)
    if (!strcmp(key, "predecode")) { h->predecode = value != 0; return 0; }   // 0: a piece's decoders start behind the chain / resolve / CRC kernels of the piece in front (round 3)
    if (!strcmp(key, "host_continuation")) { h->host_cont = value != 0; return 0; }   // 0: a stretch without findable block starts stays on the device (one wave)
    if (!strcmp(key, "far_kib")) {
        if (value < 16 || value > (1 << 20)) { h->err = "far_kib must lie in [16, 1 GiB]"; return BZQ_ERR_ARG; }
        h->far_bytes = value << 10;
        return 0;
    }
    if (!strcmp(key, "host_budget_kib")) {   // output per stay on the host before the device is asked again (default 32 MiB, doubling while it keeps handing over)
        if (value < 64 || value > (1 << 20)) { h->err = "host_budget_kib must lie in [64, 1 GiB]"; return BZQ_ERR_ARG; }
        h->host_budget = (uint64_t)value << 10; h->host_budget_min = h->host_budget;
        return 0;
    }
    if (!strcmp(key, "host_out_mib")) return (int32_t)std::min<uint64_t>(h->host_bytes_out >> 20, 0x7FFFFFFF);   // query: MiB those calls produced
    if (!strcmp(key, "host_calls")) return (int32_t)std::min<uint64_t>(h->host_calls, 0x7FFFFFFF);   // query: calls that continued on the host
    h->err = std::string("unknown option ") + key;
    return BZQ_ERR_ARG;
}
int32_t bzq_gzip_decode(bzq_gzip* h, const uint8_t* comp, uint64_t n, int32_t is_last, uint8_t* d_out, uint64_t out_capacity, uint64_t* out_bytes, int32_t* more) {
    if (!h || !out_bytes || !more || (n && !comp) || (out_capacity && !d_out)) return BZQ_ERR_ARG;
    return bzq::gz::gz_decode(h, comp, n, is_last != 0, d_out, out_capacity, out_bytes, more);
}
int32_t bzq_gzip_stage(bzq_gzip* h, const uint8_t* comp, uint64_t n) {
    if (!h || (n && !comp)) return BZQ_ERR_ARG;
    return bzq::gz::gz_stage(h, comp, n);   // (does not touch the handle's error text: that belongs to the decode thread)
}
int32_t bzq_gzip_finished(const bzq_gzip* h) { return h && h->finished ? 1 : 0; }
int32_t bzq_gzip_get_stats(const bzq_gzip* h, bzq_gzip_stats* out) {
    if (!h || !out) return BZQ_ERR_ARG;
    *out = h->stats;
    return 0;
}
const char* bzq_gzip_last_error(const bzq_gzip* h) { return h ? h->err.c_str() : "null handle"; }
void bzq_gzip_close(bzq_gzip* h) { bzq::gz::gz_free(h); }

int32_t bzq_device_alloc(bzq_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    *out = nullptr;
    const hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) { c->err = "bzq_device_alloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e); return BZQ_ERR_NOMEM; }
    return 0;
}
int32_t bzq_device_free(bzq_ctx* c, void* p) {
    if (!c) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (p) HIPCHK(c, hipFree(p));
    return 0;
}
int32_t bzq_copy_to_device(bzq_ctx* c, void* d_dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!d_dst || !src))) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    // On the ctx stream and waited for: a plain hipMemcpy from pageable memory may return once the bytes are staged, with the
    // DMA still in flight on the null stream -- which the library's (non-blocking) streams do not wait for
    if (bytes) {
        HIPCHK(c, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

int32_t bzq_submit_chunk_device(bzq_ctx* c, const uint8_t* d_data, uint64_t n, uint64_t stream_pos, int32_t is_eof) {
    if (!c || (!d_data && n)) return BZQ_ERR_ARG;
    if (((uintptr_t)d_data & 15u) != 0) { c->err = "device chunk must be 16-byte aligned"; return BZQ_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    c->shard_mode = false;
    return submit_common(c, d_data, n, stream_pos, is_eof, 0, 0, 0, 0, 10u, 0);
}

int32_t bzq_submit_chunk_host(bzq_ctx* c, const uint8_t* data, uint64_t n, uint64_t stream_pos, int32_t is_eof) {
    if (!c || (!data && n)) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (c->pending) HIPCHK(c, hipStreamSynchronize(c->stream));   // the previous chunk may still be reading c->in
    if ((rc = ensure(c, c->in, (size_t)n + 64))) return rc;
    if (n) HIPCHK(c, hipMemcpyAsync(c->in.p, data, n, hipMemcpyHostToDevice, c->stream));
    c->shard_mode = false;
    return submit_common(c, (const uint8_t*)c->in.p, n, stream_pos, is_eof, 0, 0, 0, 0, 10u, 0);
}

int32_t bzq_chunk_result(bzq_ctx* c, bzq_chunk* out) {
    if (!c || !out) return BZQ_ERR_ARG;
    if (!c->pending && !c->have_result) { c->err = "no chunk submitted"; return BZQ_ERR_ARG; }
    if (c->have_result) { *out = c->res; return (c->res.status > 0 && c->res.status != BZQ_EOF) ? c->res.status : 0; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ChunkState* h = c->h_state; // filled by the async D2H enqueued behind k_rebase
    // diagnostic snapshots of the state as the host first sees it, and again behind a re-make (query "dump_state" prints them)
    auto snap = [&](int i) { int64_t* d = c->dbg[i]; d[0] = (int64_t)h->err_struct; d[1] = h->rec_overflow; d[2] = h->P; d[3] = h->n_complete; d[4] = (int64_t)h->err_valid;
                             d[5] = (int64_t)h->err_buf; d[6] = h->P0; d[7] = h->last_nl_tile; d[8] = h->views_fallback; d[9] = h->tail_start; d[10] = h->last_record_end; d[11] = 1; };
    c->dbg[1][11] = 0; c->dbg[2][11] = 0;
    snap(0);
    if (c->ablate & 64) {   // debug: emit-kernel phase cycles (thread 0 of every workgroup)
        fprintf(stderr, "bzq phase_cycles:");
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %llu", h->phase_cycles[i]);
        fprintf(stderr, "\n");
    }
    if (h->lookback_timeout && c->cur_n > 0) {
        // never expected: the single-pass kernel gave up on a predecessor tile.  Same chunk again on
        // the two-pass kernels (no inter-workgroup waiting).
        const int why = h->lookback_timeout;   // 2: pass A's hypothesis was contradicted
        c->published = false;   // (the repeats below copy the state and the batch table back themselves)
        ChunkState fresh = *h;
        fresh.P = fresh.P0; fresh.S = fresh.S0; fresh.Q = fresh.Q0; fresh.I = fresh.I0;
        fresh.last_nl_tile = -1; fresh.rec_overflow = 0; fresh.lookback_timeout = 0; fresh.dense_tiles = 0;
        fresh.err_struct = ~0ull; fresh.err_valid = ~0ull; fresh.err_buf = ~0ull;
        *h = fresh;
        HIPCHK(c, hipMemcpyAsync(c->d_state, h, sizeof(ChunkState), hipMemcpyHostToDevice, c->stream));
        int rc;
        c->ran_single_pass = false; c->ran_stream = false; c->stream_fallbacks += 1;
        if ((rc = ensure_tile_arenas(c, c->cur_n))) return rc;
        decide_fold(c);
        if (why == 2 && c->sticky_opt) {   // pass A's hypothesis was contradicted: the next chunks of this file will be too
            c->hyp_contradictions += 1;
            c->exact_sticky_len = std::min(1024, std::max(16, 2 * c->exact_sticky_len));
            c->exact_sticky = c->exact_sticky_len;
        }
        c->exact_pass_a = true; c->used_h = false;
        rc = enqueue_passes(c, false, false);
        c->exact_pass_a = false;
        if (rc) return rc;
        hipLaunchKernelGGL(k_tail, dim3(1), dim3(BLOCK), 0, c->stream, c->cur, (int64_t)c->cur_n, c->d_state, finish_args(c, false));
        enqueue_rebase(c);
        HIPCHK(c, hipMemcpyAsync(h, c->d_state, sizeof(ChunkState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (h->views_fallback && c->cur_n > 0) {
        // views mode, a chunk of records of a few bytes: more tiles needed a big entry slot than the pool has.  Same
        // chunk again on the byte-level views kernels (two reads of the input).
        c->published = false;
        ChunkState fresh = *h;
        fresh.P = fresh.P0; fresh.S = fresh.S0; fresh.Q = fresh.Q0; fresh.I = fresh.I0;
        fresh.last_nl_tile = -1; fresh.rec_overflow = 0; fresh.views_fallback = 0; fresh.listed_tiles = 0; fresh.dense_tiles = 0;
        fresh.err_struct = ~0ull; fresh.err_valid = ~0ull; fresh.err_buf = ~0ull;
        *h = fresh;
        HIPCHK(c, hipMemcpyAsync(c->d_state, h, sizeof(ChunkState), hipMemcpyHostToDevice, c->stream));
        int rc;
        c->views_bytes_once = true;
        if ((rc = enqueue_passes(c, false, false))) return rc;
        hipLaunchKernelGGL(k_tail, dim3(1), dim3(BLOCK), 0, c->stream, c->cur, (int64_t)c->cur_n, c->d_state, finish_args(c, false));
        enqueue_rebase(c);
        HIPCHK(c, hipMemcpyAsync(h, c->d_state, sizeof(ChunkState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        snap(1);
    }
    if (h->rec_overflow) {
        // shorter records than the sizing hint assumed: re-size to the exact count, re-run
        c->published = false;
        const int64_t need = std::max<int64_t>(0, h->P >> 2) + 2;
        int rc;
        if ((rc = ensure_record_arenas(c, need + 1024))) return rc;
        ChunkState fresh = *h;
        fresh.rec_overflow = 0; fresh.err_struct = ~0ull; fresh.err_valid = ~0ull; fresh.err_buf = ~0ull;
        fresh.dense_tiles = 0;
        if (c->ran_single_pass || c->ran_stream) { fresh.last_nl_tile = -1; }
        *h = fresh;
        HIPCHK(c, hipMemcpyAsync(c->d_state, h, sizeof(ChunkState), hipMemcpyHostToDevice, c->stream));
        if (c->ran_single_pass) { if ((rc = enqueue_single_launch(c))) return rc; }
        else if (c->ran_stream) { if ((rc = enqueue_stream(c))) return rc; }
        else if ((rc = enqueue_passes(c, true, false))) return rc;
        enqueue_rebase(c);
        HIPCHK(c, hipMemcpyAsync(h, c->d_state, sizeof(ChunkState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        snap(2);
        if (h->rec_overflow) { c->err = "record arrays still too small after re-size"; return BZQ_ERR_NOMEM; }
    }
    const int64_t n = (int64_t)c->cur_n;
    const int64_t lines = h->P;                       // P0 + total newlines
    int64_t n_complete = lines > 0 ? (lines >> 2) : 0; // records with all four newlines
    const int tail_phase = (int)(lines & 3);
    const int64_t batch = c->cfg.batch_size;
    const bool growth = c->cfg.buffer_growth_enabled != 0;
    const int64_t len_limit = growth ? c->cfg.buffer_max_capacity : c->cfg.buffer_capacity;
    int64_t consumed = n_complete > 0 ? h->last_record_end + 1 : c->cur_first_header;

    bzq_chunk r{};
    r.n_bytes = c->cur_n;
    r.total_newlines = (uint64_t)(h->P - h->P0);
    r.tail_phase = tail_phase;
    r.error_record = -1;
    r.status = BZQ_OK;
    c->term_phase = 0; c->term_cap = c->cfg.buffer_capacity;

    if (c->records_before >= 0 && !c->shard_mode && !c->cur_is_eof) {   // (a stream's last chunk walks its own records below, if it has to)
        int rc2;
        if ((rc2 = follow_stage(c, n_complete))) return rc2;
    }
    auto key_rec = [](u64 k) { return (int64_t)(k >> 3); };
    // first failing record among the complete ones; same record: buffer < structure < validation
    u64 best = ~0ull;
    if (h->err_buf != ~0ull && key_rec(h->err_buf) < n_complete) best = std::min(best, h->err_buf);
    if (h->err_struct != ~0ull && key_rec(h->err_struct) < n_complete) best = std::min(best, h->err_struct);
    if (h->err_valid != ~0ull && key_rec(h->err_valid) < n_complete) best = std::min(best, h->err_valid);
    int64_t n_records = n_complete;
    bool accept_last = false;
    if (best != ~0ull) {
        n_records = key_rec(best);
        int code = (int)(best & 7);
        if (code == 0) { code = growth ? BZQ_BUFFER_AT_MAX : BZQ_BUFFER_EXCEEDED; c->term_cap = len_limit; }
        r.status = code;
        r.error_record = n_records;
    } else if (c->cur_is_eof && !c->shard_mode) {
        if (consumed >= n) {
            r.status = BZQ_EOF;
        } else {
            // the records the window has passed over: this chunk's -- behind, for a stream that came in several chunks, the window
            // state the follower carried over every record handed out before it (stream offsets)
            std::vector<int64_t> re((size_t)n_complete);
            if (n_complete) HIPCHK(c, hipMemcpy(re.data(), c->o().rec_end.p, (size_t)n_complete * 8, hipMemcpyDeviceToHost));
            int ph = 0; int64_t cap = c->cfg.buffer_capacity;
            int code;
            if (c->records_before > 0 && c->follow_on && c->follow_records == c->records_before) {
                Window sw = c->follow;
                int64_t head = c->follow_head;
                const int64_t N = (int64_t)c->cur_stream_pos + (int64_t)n;
                window_walk(sw, head, re.data(), re.size(), (int64_t)c->cur_stream_pos, c->cfg);
                sw.N = N;
                if (sw.end > N) sw.end = N;
                code = window_classify(sw, (int64_t)c->cur_stream_pos + consumed, c->cfg, tail_phase, h->tail_nonblank != 0, &accept_last, &ph, &cap);
            } else {
                code = classify_tail(c, re, (int64_t)n, c->cur_first_header, consumed, tail_phase, h->tail_nonblank != 0, &accept_last, &ph, &cap);
            }
            c->term_phase = ph; c->term_cap = cap;
            if (accept_last) {
                // last record without trailing newline (Q4): structure check skipped, validation still applies
                if (n_complete + 1 > c->o().rec_cap) {
                    c->err = "record arrays too small for the unterminated last record";
                    return BZQ_ERR_NOMEM;
                }
                launch_fix_last(c, n_complete, n, batch);
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (h->err_valid != ~0ull && key_rec(h->err_valid) == n_complete) {
                    // ... and fails it: the record is not delivered, the stream stops behind the last complete one
                    r.status = (int)(h->err_valid & 7);
                    r.error_record = n_complete;
                } else {
                    r.status = BZQ_EOF;
                    n_records = n_complete + 1;
                    consumed = n;
                }
            } else {
                r.status = code;
                r.error_record = (code == BZQ_EOF) ? -1 : n_complete;
            }
        }
    } else if (c->cur_is_eof && c->shard_mode) {
        // last shard.  Which outcome the reference gives for trailing bytes that are not a record depends on where its
        // BufferedReader window sits, i.e. on every record since the stream's first byte:
        //   tail_mode 1  (bzq_shard_stitch, first look): the decision is deferred -- the protocol walks the window through
        //                the ranks' records in rank order and comes back with it (tail_mode 2);
        //   tail_mode 0  (bare bzq_submit_shard): the outcome the reference gives when its window holds the tail completely.
        int code = BZQ_OK;
        bool acc = false;
        if (consumed >= n) code = BZQ_EOF;
        else if (c->tail_mode == 1) { c->tail_pending = true; code = BZQ_OK; }
        else if (c->tail_mode == 2) { code = c->tail_code; acc = c->tail_accept; c->term_phase = c->tail_phase_dec; c->term_cap = c->tail_cap_dec; }
        else if (tail_phase == 3 && h->tail_nonblank) acc = true;
        else if (tail_phase == 3) code = BZQ_OTHER;
        else { code = BZQ_UNEXPECTED_EOF; c->term_phase = tail_phase; }
        if (acc) {
            if (n_complete + 1 > c->o().rec_cap) { c->err = "record arrays too small"; return BZQ_ERR_NOMEM; }
            launch_fix_last(c, n_complete, n, batch);
            HIPCHK(c, hipStreamSynchronize(c->stream));
            accept_last = true;
            if (h->err_valid != ~0ull && key_rec(h->err_valid) == n_complete) {
                r.status = (int)(h->err_valid & 7); r.error_record = n_complete;
            } else { r.status = BZQ_EOF; n_records = n_complete + 1; consumed = n; }
        } else if (code == BZQ_EOF || (code == BZQ_OK && c->tail_pending)) {
            r.status = code;
        } else { r.status = code; r.error_record = n_complete; }
    }

    r.n_records = (uint64_t)n_records;
    if (r.status > 0 && r.status != BZQ_EOF && r.error_record >= 0 && r.error_record < n_complete && n_records < n_complete) {
        // consumed = end of the last delivered record
        if (n_records > 0) {
            int64_t le = 0;
            HIPCHK(c, hipMemcpy(&le, (const int64_t*)c->o().rec_end.p + (n_records - 1), 8, hipMemcpyDeviceToHost));
            consumed = le + 1;
        } else consumed = c->cur_first_header;
    }
    r.bytes_consumed = (uint64_t)consumed;
    if (n_records > 0 && !c->cfg.views_only) {
        int64_t e2[2] = {h->last_ends, h->last_id_ends};
        if (n_records != n_complete) { // truncated by an error, or extended by the unterminated last record
            int rc3;
            if ((rc3 = cum_pair(c, n_records - 1, e2))) return rc3;
        }
        r.qual_bytes = (uint64_t)e2[0];
        r.seq_bytes = accept_last && r.status == BZQ_EOF ? (uint64_t)h->S : (uint64_t)e2[0];
        r.id_bytes = (uint64_t)e2[1];
    }
    if (!c->cfg.views_only) {
        r.d_seq = (const uint8_t*)c->o().seq.p; r.d_qual = (const uint8_t*)c->o().qual.p; r.d_id = (const uint8_t*)c->o().id.p;
        // chunk-cumulative arrays: filled with the chunk only on the k_rebase path; otherwise on demand (bzq_chunk_cumulative_ends)
        if (!c->fold) { r.d_ends = (const int64_t*)c->o().ends.p; r.d_id_ends = (const int64_t*)c->o().id_ends.p; }
        r.d_batch_ends = (const int64_t*)c->o().b_ends.p; r.d_batch_id_ends = (const int64_t*)c->o().b_id_ends.p;
    } else {
        r.d_id_start = (const int64_t*)c->o().id_start.p; r.d_id_len = (const int32_t*)c->o().id_len.p;
    }
    r.d_record_end = (const int64_t*)c->o().rec_end.p;
    if (c->cfg.emit_offsets || c->cfg.views_only) {
        r.d_header_start = (const int64_t*)c->o().off[0].p; r.d_seq_start = (const int64_t*)c->o().off[1].p;
        r.d_sep_start = (const int64_t*)c->o().off[2].p; r.d_qual_start = (const int64_t*)c->o().off[3].p;
    }
    float ms0 = 0.f, ms1 = 0.f;
    if (hipEventElapsedTime(&ms0, c->ev[0], c->ev[3]) != hipSuccess) ms0 = 0.f;
    // what follows the last emit (k_tail, k_rebase / k_finish): known apart only with the marks of option timing_detail
    if (c->timing_detail && !c->ev_detail.empty() && hipEventElapsedTime(&ms1, c->ev_detail.back(), c->ev[3]) != hipSuccess) ms1 = 0.f;
    r.ms_total = ms0;
    r.ms_rebase = ms1;
    if (c->shard_mode) { r.ms_total += c->ms_scan_shard; r.ms_aggregate += c->ms_scan_shard; }   // pass A ran in bzq_shard_scan
    if (c->timing_detail && (c->ran_single_pass || c->ran_stream) && c->ev_detail.size() >= 2) {
        float d = 0;
        if (hipEventElapsedTime(&d, c->ev_detail[c->ev_detail.size() - 2], c->ev_detail[c->ev_detail.size() - 1]) == hipSuccess) r.ms_emit = d;
    } else if (c->timing_detail && c->ev_detail.size() >= 4) {
        for (size_t i = 0; i + 3 < c->ev_detail.size(); i += 4) {
            float a = 0, b = 0, d = 0;
            if (hipEventElapsedTime(&a, c->ev_detail[i], c->ev_detail[i + 1]) != hipSuccess) a = 0;
            if (hipEventElapsedTime(&b, c->ev_detail[i + 1], c->ev_detail[i + 2]) != hipSuccess) b = 0;
            if (hipEventElapsedTime(&d, c->ev_detail[i + 2], c->ev_detail[i + 3]) != hipSuccess) d = 0;
            r.ms_aggregate += a; r.ms_scan += b; r.ms_emit += d;
        }
    }
    // the batch-boundary table of the complete records came back with the state (async copy behind k_rebase)
    {
        OutSet& o = c->o();
        const int64_t nb = (n_complete + batch - 1) / batch;
        o.h_bb_records = (!c->cfg.views_only && o.h_bb && nb > 0 && nb <= o.h_bb_batches) ? n_complete : 0;
    }
    r.n_passes = (uint32_t)c->n_passes;
    r.chunk_serial = (uint32_t)c->n_submits;
    c->last_dense_tiles = (int64_t)h->dense_tiles;
    c->res = r;
    { OutSet& o = c->o(); o.fold = c->fold; o.cum_valid = !c->fold; o.parsed = !c->cfg.views_only; o.batch = std::max<int64_t>(1, c->cfg.batch_size); o.n_records = (int64_t)r.n_records;
      o.serial = r.chunk_serial; }
    c->pending = false; c->have_result = true;
    c->res_set = c->cur_set; c->res_alive = true; c->res_cur = c->cur;
    *out = r;
    return (r.status > 0 && r.status != BZQ_EOF) ? r.status : 0;
}

int32_t bzq_batch_view(bzq_ctx* c, uint64_t first_record, uint32_t max_records, bzq_device_batch* out) {
    if (!c || !out || !(c->have_result || c->res_alive)) { if (c) c->err = "bzq_batch_view: no parsed chunk (or two chunks have been submitted since its result was taken)"; return BZQ_ERR_ARG; }
    if (max_records == 0) { c->err = "bzq_batch_view: max_records must be > 0"; return BZQ_ERR_ARG; }
    if (c->cfg.views_only) { c->err = "bzq_batch_view: the ctx is in views mode (no columns); use bzq_views"; return BZQ_ERR_ARG; }
    const uint64_t bs = (uint64_t)c->cfg.batch_size;
    memset(out, 0, sizeof(*out));
    out->quality_offset = 33; // parser.mojo:243 builds FastqBatch(batch_size=limit): default offset
    out->first_record = first_record;
    if (first_record >= c->res.n_records) return 0; // empty batch: the iterator stops (parser.mojo:727-729)
    const uint64_t nrec = std::min<uint64_t>(max_records, c->res.n_records - first_record);
    int64_t base[2] = {0, 0}, last[2] = {0, 0};
    OutSet& os = c->out[c->res_set];   // (the set of the chunk `res` describes: the current one, or the one before while the next is in flight)
    const uint64_t lastrec = first_record + nrec - 1;
    // batch aligned inside the complete records: both ends come from the host copy of the batch-boundary table
    const bool cached = os.h_bb_records > 0 && first_record % bs == 0 && lastrec < (uint64_t)os.h_bb_records &&
                        ((lastrec + 1) % bs == 0 || lastrec + 1 == (uint64_t)os.h_bb_records) && nrec <= bs;
    if (!cached || first_record % bs != 0 || max_records > bs) HIPCHK(c, hipSetDevice(c->device));
    if (cached) {
        const uint64_t k = first_record / bs;
        if (k > 0) { base[0] = os.h_bb[2 * (k - 1)]; base[1] = os.h_bb[2 * (k - 1) + 1]; }
        last[0] = os.h_bb[2 * k]; last[1] = os.h_bb[2 * k + 1];
    } else {
        int rc4;
        if (first_record > 0 && (rc4 = cum_pair(c, os, os.fold, os.cum_valid, (int64_t)first_record - 1, base))) return rc4;
        if ((rc4 = cum_pair(c, os, os.fold, os.cum_valid, (int64_t)lastrec, last))) return rc4;
    }
    out->num_records = (int64_t)nrec;
    out->seq_len = last[0] - base[0];
    out->total_id_bytes = last[1] - base[1];
    // the sequence column runs in step with the quality column (equal lengths are checked per record) except through
    // an accepted unterminated last record: a batch that reaches the chunk's last record ends where the column ends
    out->sequence_bytes = (first_record + nrec == c->res.n_records ? (int64_t)c->res.seq_bytes : last[0]) - base[0];
    out->qual_buffer = c->res.d_qual + base[0];
    out->sequence_buffer = c->res.d_seq + base[0];
    out->id_buffer = c->res.d_id + base[1];
    if (first_record % bs == 0 && max_records <= bs) {
        // batch aligned: the per-batch ends were produced with the chunk (k_rebase), zero copy
        out->ends = c->res.d_batch_ends + first_record;
        out->id_ends = c->res.d_batch_id_ends + first_record;
    } else {
        // its own storage: a view handed out earlier keeps its ends (they stay valid as long as the chunk's columns)
        int rc;
        void* ve = nullptr;
        if ((rc = view_alloc(c, os, (size_t)nrec * 16, &ve))) return rc;
        int64_t* e = (int64_t*)ve;
        // (on the ctx stream: behind the parse of the next chunk if one is in flight -- it writes the OTHER set)
        if (os.fold && !os.cum_valid)
            hipLaunchKernelGGL(k_rebase_range_b, dim3((unsigned)((nrec + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, c->res.d_batch_ends,
                               c->res.d_batch_id_ends, (const int64_t*)os.bb.p, (int64_t)bs, (int64_t)first_record, (int64_t)nrec, e, e + nrec);
        else
            hipLaunchKernelGGL(k_rebase_range, dim3((unsigned)((nrec + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream,
                               (const int64_t*)os.ends.p, (const int64_t*)os.id_ends.p, (int64_t)first_record, (int64_t)nrec, e, e + nrec);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        out->ends = e;
        out->id_ends = e + nrec;
    }
    return 0;
}

int32_t bzq_batches(bzq_ctx* c, uint32_t max_records, bzq_device_batch* out, uint64_t cap, uint64_t* n_out) {
    if (!c || !n_out || (cap && !out) || !(c->have_result || c->res_alive)) { if (c) c->err = "bzq_batches: no parsed chunk (or two chunks have been submitted since its result was taken)"; return BZQ_ERR_ARG; }
    if (max_records == 0) { c->err = "bzq_batches: max_records must be > 0"; return BZQ_ERR_ARG; }
    const uint64_t n = c->res.n_records, nb = (n + max_records - 1) / max_records;
    *n_out = nb;
    for (uint64_t k = 0; k < nb && k < cap; ++k) {
        const int32_t rc = bzq_batch_view(c, k * (uint64_t)max_records, max_records, &out[k]);
        if (rc < 0) return rc;
    }
    return 0;
}

int32_t bzq_chunk_cumulative_ends(bzq_ctx* c, bzq_chunk* inout) {
    if (!c) return BZQ_ERR_ARG;
    if (c->cfg.views_only) { c->err = "bzq_chunk_cumulative_ends: the ctx is in views mode (no columns)"; return BZQ_ERR_ARG; }
    // which chunk: the one `inout` describes -- its per-batch arrays live in exactly one output set -- else the current one.  The
    // set keeps what the derivation needs (OutSet::fold, batch, n_records), so the chunk BEFORE the current one is served too, as
    // long as it is alive (until the second submit after its own).
    // A set's pointers outlive the chunk (the arrays are reused): the chunk is recognised by its pointers AND its serial, so a
    // bzq_chunk from two submits ago is refused instead of being served the newer chunk's ends in its name.  The set of a
    // chunk whose successor is still in flight (no result taken yet) is served too: OutSet::parsed, not "the ctx has a result".
    OutSet* os = nullptr;
    if (inout && inout->d_batch_ends) {
        for (OutSet& cand : c->out)
            if (cand.parsed && (const int64_t*)cand.b_ends.p == inout->d_batch_ends && cand.serial == inout->chunk_serial) os = &cand;
        if (!os) { c->err = "bzq_chunk_cumulative_ends: the chunk's arrays are no longer alive (two chunks have been submitted since)"; return BZQ_ERR_ARG; }
    } else {
        if (!c->have_result) { c->err = "bzq_chunk_cumulative_ends: no parsed chunk"; return BZQ_ERR_ARG; }
        os = &c->o();
    }
    const bool current = os == &c->o() && c->have_result;
    if (os->fold && !os->cum_valid && os->n_records > 0) {
        HIPCHK(c, hipSetDevice(c->device));
        const int64_t n = os->n_records;
        const unsigned grid = (unsigned)std::min<int64_t>((n + BLOCK - 1) / BLOCK, (int64_t)c->num_cu * 16);
        // (on the ctx stream: behind the parse that may be running there, which writes the OTHER set)
        hipLaunchKernelGGL(k_cumulate, dim3(grid), dim3(BLOCK), 0, c->stream, (const int64_t*)os->b_ends.p, (const int64_t*)os->b_id_ends.p,
                           (const int64_t*)os->bb.p, os->batch, n, (int64_t*)os->ends.p, (int64_t*)os->id_ends.p);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    os->cum_valid = true;
    if (current) { c->cum_valid = true; c->res.d_ends = (const int64_t*)os->ends.p; c->res.d_id_ends = (const int64_t*)os->id_ends.p; }
    if (inout) { inout->d_ends = (const int64_t*)os->ends.p; inout->d_id_ends = (const int64_t*)os->id_ends.p; }
    return 0;
}

int32_t bzq_views(bzq_ctx* c, uint64_t first_record, uint32_t max_records, bzq_device_views* out) {
    if (!c || !out || !(c->have_result || c->res_alive)) { if (c) c->err = "bzq_views: no parsed chunk (or two chunks have been submitted since its result was taken)"; return BZQ_ERR_ARG; }
    if (!c->cfg.views_only) { c->err = "bzq_views: the ctx is not in views mode (config.views_only)"; return BZQ_ERR_ARG; }
    // views point INTO the chunk: served behind the next submit only where the chunk's bytes are the caller's (bzq_submit_chunk_device);
    // a host submit's bytes sit in the ctx's staging buffer, which the next host submit overwrites
    if (!c->have_result && c->res_cur && c->res_cur == (const uint8_t*)c->in.p) {
        c->err = "bzq_views: the chunk was submitted from host memory and the next chunk has been submitted since (its bytes on the device are being overwritten)";
        return BZQ_ERR_ARG;
    }
    memset(out, 0, sizeof(*out));
    out->first_record = first_record;
    out->chunk = c->res_cur;
    if (first_record >= c->res.n_records) return 0;
    out->num_records = (int64_t)std::min<uint64_t>(max_records, c->res.n_records - first_record);
    out->header_start = c->res.d_header_start + first_record; out->seq_start = c->res.d_seq_start + first_record;
    out->sep_start = c->res.d_sep_start + first_record; out->qual_start = c->res.d_qual_start + first_record;
    out->record_end = c->res.d_record_end + first_record;
    out->id_start = c->res.d_id_start + first_record; out->id_len = c->res.d_id_len + first_record;
    return 0;
}

int32_t bzq_batch_to_host(bzq_ctx* c, const bzq_device_batch* b, bzq_host_batch* out) {
    if (!c || !b || !out) return BZQ_ERR_ARG;
    out->num_records = b->num_records;
    out->quality_offset = b->quality_offset;
    if (b->num_records == 0) return 0;
    HIPCHK(c, hipMemcpy(out->quality_bytes, b->qual_buffer, (size_t)b->seq_len, hipMemcpyDeviceToHost));
    if (b->sequence_bytes > 0) HIPCHK(c, hipMemcpy(out->sequence_bytes, b->sequence_buffer, (size_t)b->sequence_bytes, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out->id_bytes, b->id_buffer, (size_t)b->total_id_bytes, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out->ends, b->ends, (size_t)b->num_records * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out->id_ends, b->id_ends, (size_t)b->num_records * 8, hipMemcpyDeviceToHost));
    return 0;
}

int32_t bzq_copy_to_host(bzq_ctx* c, void* dst, const void* d_src, size_t bytes) {
    if (!c || (!dst && bytes)) return BZQ_ERR_ARG;
    if (bytes) HIPCHK(c, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int64_t bzq_format_error(bzq_ctx* c, uint64_t records_before, char* buf, size_t cap) {
    if (!c || !c->have_result) return -1;
    const bzq_chunk& r = c->res;
    std::string s;
    const int code = r.status;
    if (code == BZQ_OK) s = "";
    else if (code == BZQ_EOF) s = "EOF"; // CONSTS.mojo:19
    else if (code == BZQ_OTHER) s = "";
    else if (code == BZQ_UNEXPECTED_EOF) {
        s = "Unexpected end of file in FASTQ record at phase " + std::to_string(c->term_phase); // parser.mojo:295-298
    } else if (code == BZQ_BUFFER_EXCEEDED) {
        s = "FASTQ record exceeds buffer capacity (" + std::to_string(c->term_cap) +
            " bytes). Enable buffer growth or increase buffer_capacity."; // parser.mojo:299-304
    } else if (code == BZQ_BUFFER_AT_MAX) {
        s = "FASTQ record exceeds maximum buffer capacity (" + std::to_string(c->cfg.buffer_max_capacity) +
            " bytes). Enable buffer growth or increase max_capacity."; // parser.mojo:305-309
    } else {
        // the failing record's bytes: [start, record_end]
        const int64_t rec = r.error_record;
        int64_t start = c->cur_first_header, end = 0;
        // a failed copy must not format garbage into the text: the call then fails as a whole (-1, text in bzq_last_error)
        auto fetch_fail = [&](hipError_t e) { c->err = std::string("bzq_format_error: device read failed: ") + hipGetErrorString(e); return (int64_t)-1; };
        hipError_t ce;
        if (rec > 0) {
            if ((ce = hipMemcpy(&start, r.d_record_end + (rec - 1), 8, hipMemcpyDeviceToHost)) != hipSuccess) return fetch_fail(ce);
            start += 1;
        }
        if ((uint64_t)rec < (uint64_t)(c->h_state->P >> 2)) {
            if ((ce = hipMemcpy(&end, r.d_record_end + rec, 8, hipMemcpyDeviceToHost)) != hipSuccess) return fetch_fail(ce);
        } else end = (int64_t)c->cur_n - 1; // unterminated last record
        int64_t len = end - start + 1;
        if (len < 0) len = 0;
        const int64_t fetch = std::min<int64_t>(len, 1 << 20);
        std::vector<uint8_t> raw((size_t)fetch);
        if (fetch && (ce = hipMemcpy(raw.data(), c->cur + start, (size_t)fetch, hipMemcpyDeviceToHost)) != hipSuccess) return fetch_fail(ce);
        const long long recno = (long long)(records_before + (uint64_t)rec + 1);
        s = message_for_code(code);
        if (code <= BZQ_SEQ_QUAL_LEN_MISMATCH) {
            // ParseError.write_to (errors.mojo:178-192) with parser.mojo:332-338 numbering
            sb_put(s, "\n  Record number: ", recno);
            sb_put(s, "\n  Line number: ", 4 * (recno - 1) + 1);
            const long long pos = (long long)(c->cur_stream_pos + (uint64_t)start);
            if (pos > 0) sb_put(s, "\n  File position: ", pos);
            const int64_t sn = std::min<int64_t>(std::min<int64_t>(len, 200), fetch); // utils.mojo:435-445
            if (sn > 0) { s += "\n  Record snippet: "; s.append((const char*)raw.data(), (size_t)sn); }
        } else {
            // ValidationError.write_to (errors.mojo:223-234); snippet = id "\n" sequence prefix (parser.mojo:597-610)
            sb_put(s, "\n  Record number: ", recno);
            int64_t nl[3] = {-1, -1, -1}; int k = 0;
            for (int64_t i = 0; i < fetch && k < 3; ++i) if (raw[(size_t)i] == 10) nl[k++] = i;
            std::string snip;
            if (k >= 1) {
                int64_t a = 1, b = nl[0];
                auto sp = [](uint8_t ch) { return ch <= 32 && ((0x170003E00ull >> ch) & 1ull); };
                if (b > a && (sp(raw[(size_t)a]) || sp(raw[(size_t)(b - 1)]))) {
                    while (a < b && sp(raw[(size_t)a])) ++a;
                    while (b > a && sp(raw[(size_t)(b - 1)])) --b;
                }
                if (b > a) {
                    snip.append((const char*)raw.data() + a, (size_t)(b - a));
                    if (snip.size() < 200) snip += "\n";
                }
                const int64_t s0 = nl[0] + 1, s1 = k >= 2 ? nl[1] : fetch;
                if (snip.size() < 200 && s1 > s0) {
                    const int64_t take = std::min<int64_t>(s1 - s0, 200 - (int64_t)snip.size());
                    snip.append((const char*)raw.data() + s0, (size_t)take);
                }
                if (snip.size() > 200) snip = snip.substr(0, 197) + "...";
            }
            if (!snip.empty()) { s += "\n  Record snippet: "; s += snip; }
        }
    }
    if (buf && cap) {
        const size_t nbytes = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), nbytes);
        buf[nbytes] = 0;
    }
    return (int64_t)s.size();
}

// The scan of a shard in two halves, so that bzq_shard_stitch can put the summary all-gather on the stream between them:
// enqueue = pass A + tile scan + first newlines + the state's copy back (ev[4] .. ev[5] time the kernels), finish = read
// the summary out of the pinned state once the stream has been synchronised.
static int shard_scan_enqueue(bzq_ctx* c, const uint8_t* d_data, uint64_t n) {
    int rc;
    if (c->pending) HIPCHK(c, hipStreamSynchronize(c->stream));   // h_state is reused
    if ((rc = ensure_tile_arenas(c, n + (64u << 20)))) return rc; // room for the halo tiles too
    ChunkState* h = c->h_state;
    memset(h, 0, sizeof(*h));
    h->last_nl_tile = -1;
    h->err_struct = ~0ull; h->err_valid = ~0ull; h->err_buf = ~0ull;
    for (int i = 0; i < 4; ++i) h->first_nl[i] = -1;
    h->edge_first = 10; h->edge_last = 10;
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(c->d_state, h, sizeof(ChunkState), hipMemcpyHostToDevice, c->stream));
        const int64_t nt = tiles_for(n);
        AggArgs a{d_data, (int64_t)n, 10u, 0, nt, (uint32_t*)c->tile_c.p, (u64*)c->tile_a.p, (u64*)c->tile_idc.p, walk_limit_of(c), (u64*)c->tile_last.p};
        c->used_h = c->pass_a_h != 0 && c->exact_sticky == 0;
        if (c->used_h) hipLaunchKernelGGL(k_tile_aggregate_h, dim3((unsigned)nt), dim3(BLOCK), 0, c->stream, a);
        else hipLaunchKernelGGL(k_tile_aggregate2, dim3((unsigned)nt), dim3(BLOCK), 0, c->stream, a);
        launch_scan(c, 0, nt, 0);
        hipLaunchKernelGGL(k_first_newlines, dim3(1), dim3(BLOCK), 0, c->stream, d_data, (int64_t)n, c->d_state);
    }
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    // the whole summary comes back with the state: one copy into pinned memory
    if (n > 0) HIPCHK(c, hipMemcpyAsync(h, c->d_state, sizeof(ChunkState), hipMemcpyDeviceToHost, c->stream));
    return 0;
}
static void shard_scan_finish(bzq_ctx* c, const uint8_t* d_data, uint64_t n, bzq_shard_summary* out) {
    const ChunkState* h = c->h_state;
    memset(out, 0, sizeof(*out));
    out->n_bytes = n;
    out->n_newlines = n ? (uint64_t)h->P : 0;
    for (int i = 0; i < 4; ++i) out->first_nl[i] = n ? h->first_nl[i] : -1;
    out->first_byte = n ? h->edge_first : 10; out->last_byte = n ? h->edge_last : 10;
    c->agg_ptr = d_data; c->agg_n = n;
    c->ms_scan_shard = 0.f;
    if (n && hipEventElapsedTime(&c->ms_scan_shard, c->ev[4], c->ev[5]) != hipSuccess) c->ms_scan_shard = 0.f;
}

int32_t bzq_shard_scan(bzq_ctx* c, const uint8_t* d_data, uint64_t n, bzq_shard_summary* out) {
    if (!c || !out || (!d_data && n)) return BZQ_ERR_ARG;
    if (((uintptr_t)d_data & 15u) != 0) { c->err = "device shard must be 16-byte aligned"; return BZQ_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = shard_scan_enqueue(c, d_data, n))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    shard_scan_finish(c, d_data, n, out);
    return 0;
}

int32_t bzq_shard_head_bytes(const bzq_shard_summary* s, uint64_t lines_before, uint8_t prev_last_byte,
                             uint64_t* head_bytes) {
    if (!s || !head_bytes) return BZQ_ERR_ARG;
    const int p0 = (int)(lines_before & 3);
    if (prev_last_byte == 10 && p0 == 0) { *head_bytes = 0; return 0; }
    const int k = 4 - p0; // lines left of the record that started in an earlier shard
    if (s->first_nl[k - 1] < 0) return BZQ_ERR_ARG; // record longer than a whole shard
    *head_bytes = (uint64_t)s->first_nl[k - 1] + 1;
    return 0;
}

int32_t bzq_submit_shard(bzq_ctx* c, const uint8_t* d_data, uint64_t n, uint64_t halo_bytes, uint64_t lines_before,
                         uint8_t prev_last_byte, uint64_t stream_pos, int32_t is_last_shard) {
    if (!c || (!d_data && n)) return BZQ_ERR_ARG;
    if (((uintptr_t)d_data & 15u) != 0) { c->err = "device shard must be 16-byte aligned"; return BZQ_ERR_ARG; }
    if (c->cfg.views_only) { c->err = "bzq_submit_shard: views mode is a single-chunk mode (shards deliver batch columns)"; return BZQ_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const bool reuse = (c->agg_ptr == d_data && c->agg_n == n && n >= (uint64_t)TILE &&
                        tiles_for(n + halo_bytes) + 1 <= c->tile_cap);
    int64_t first_nl[4] = {-1, -1, -1, -1};
    if (c->agg_ptr == d_data && c->agg_n == n) for (int i = 0; i < 4; ++i) first_nl[i] = c->h_state->first_nl[i];
    else { c->err = "bzq_submit_shard must follow bzq_shard_scan of the same shard"; return BZQ_ERR_ARG; }
    const int p0 = (int)(lines_before & 3);
    int head_lines = 0;
    int64_t first_header = 0;
    if (!(prev_last_byte == 10 && p0 == 0)) {
        head_lines = 4 - p0;
        if (first_nl[head_lines - 1] < 0) { c->err = "a record spans more than one whole shard"; return BZQ_ERR_ARG; }
        first_header = first_nl[head_lines - 1] + 1;
    }
    c->shard_mode = true;
    c->head_lines = head_lines;
    const int rc = submit_common(c, d_data, n + halo_bytes, stream_pos, is_last_shard, -(int64_t)head_lines, 0, 0, 0,
                                 prev_last_byte, first_header, first_nl, head_lines, reuse);
    c->agg_ptr = nullptr; c->agg_n = 0;
    return rc;
}

int32_t bzq_generate_synthetic_device_var(bzq_ctx* c, int64_t num_reads, int64_t first, int64_t count, int32_t min_len,
                                          int32_t max_len, int32_t min_phred, int32_t max_phred, const char* schema,
                                          uint8_t* d_out, uint64_t cap, uint64_t* out_bytes) {
    if (!c || num_reads <= 0 || first < 0 || count < 0 || first + count > num_reads || min_len < 0 || max_len < min_len ||
        min_phred < 0 || max_phred < min_phred)
        return BZQ_ERR_ARG;
    int nd = 1;
    if (num_reads > 1) nd = (int)std::to_string(num_reads - 1).size(); // utils.mojo:880-882
    const int64_t fixed = 6 + nd + 1 + 2 + 2;
    const int64_t range = (int64_t)max_len - min_len + 1;
    // lengths min_len + ((31 i + 7) mod range) repeat with period range / gcd(31, range): prefix sums of one period
    int64_t period = 1;
    std::vector<int64_t> prefix;
    uint64_t total;
    if (range > 1) {
        period = range / std::gcd<int64_t>(31, range);
        prefix.resize((size_t)period + 1);
        prefix[0] = 0;
        for (int64_t r = 0; r < period; ++r) prefix[(size_t)r + 1] = prefix[(size_t)r] + min_len + (int64_t)(((uint64_t)r * 31u + 7u) % (uint64_t)range);
        auto sum_to = [&](int64_t i) { return (i / period) * prefix[(size_t)period] + prefix[(size_t)(i % period)]; };
        total = (uint64_t)(count * fixed + 2 * (sum_to(first + count) - sum_to(first)));
    } else {
        total = (uint64_t)((fixed + 2 * (int64_t)min_len) * count);
    }
    if (out_bytes) *out_bytes = total;
    if (!d_out) return 0;
    if (cap < total) { c->err = "bzq_generate_synthetic_device: output buffer too small"; return BZQ_ERR_ARG; }
    uint8_t lo, up, off;
    (void)bzq_schema_from_name(schema ? schema : "generic", &lo, &up, &off);
    HIPCHK(c, hipSetDevice(c->device));
    int64_t* d_prefix = nullptr;
    if (range > 1) {
        int rc;
        if ((rc = ensure(c, c->gen_prefix, prefix.size() * 8))) return rc;
        d_prefix = (int64_t*)c->gen_prefix.p;
        HIPCHK(c, hipMemcpyAsync(d_prefix, prefix.data(), prefix.size() * 8, hipMemcpyHostToDevice, c->stream));
    }
    GenArgs g{d_out, first, count, num_reads, min_len, nd, min_phred, max_phred, off, lo, up, range, period, d_prefix};
    if (count > 0)
        hipLaunchKernelGGL(k_generate, dim3((unsigned)((count + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, g);
    HIPCHK(c, hipStreamSynchronize(c->stream));   // `prefix` must outlive the copy
    return 0;
}

int32_t bzq_generate_synthetic_device(bzq_ctx* c, int64_t num_reads, int64_t first, int64_t count, int32_t read_len,
                                      int32_t min_phred, int32_t max_phred, const char* schema, uint8_t* d_out,
                                      uint64_t cap, uint64_t* out_bytes) {
    return bzq_generate_synthetic_device_var(c, num_reads, first, count, read_len, read_len, min_phred, max_phred, schema,
                                             d_out, cap, out_bytes);
}

#include "bzq_comm.hpp"
namespace { double comm_timeout_s(const bzq_ctx* c) { return (double)c->comm_timeout_ms / 1e3; } }

// ---- host ingest pipeline (bzq_ingest.hpp) ---------------------------------------------------------------------

// shared by the FASTQ and the FASTA ingest: file, pinned + device double buffers, compression sniffing, producer thread
// CPUs of the NUMA node the GPU hangs off (sysfs), so that the reader threads fill the pinned buffers from the near socket
static void gpu_numa_cpus(int device, int* node_out, std::vector<int>& cpus) {
    *node_out = -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r");
    int node = -1;
    if (!f || fscanf(f, "%d", &node) != 1) node = -1;
    if (f) fclose(f);
    if (node < 0) return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return;
    int a, b;
    for (;;) {   // "0-31,64-95"
        if (fscanf(f, "%d", &a) != 1) break;
        b = a;
        int ch = fgetc(f);
        if (ch == '-') { if (fscanf(f, "%d", &b) != 1) break; ch = fgetc(f); }
        for (int c = a; c <= b; ++c) cpus.push_back(c);
        if (ch != ',') break;
    }
    fclose(f);
    if (!cpus.empty()) *node_out = node;
}

static int32_t ingest_open_common(int device, std::string& err, const char* who, const char* path, uint64_t chunk_bytes,
                                  int32_t n_threads, bzq_ingest** out, int direct = 0, int numa = 1, int gpu_inflate = 1) {
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) { err = std::string(who) + ": hipSetDevice failed"; return BZQ_ERR_HIP; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { err = std::string(who) + ": cannot open " + path; return BZQ_ERR_IO; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        err = std::string(who) + ": not a regular file: " + path;
        return BZQ_ERR_IO;
    }
    bzq_ingest* g = new bzq_ingest();
    g->device = device; g->fd = fd; g->file_size = (uint64_t)st.st_size;
    g->chunk_bytes = chunk_bytes ? ((chunk_bytes + 4095) & ~4095ull) : (256ull << 20);
    g->reserve = std::max<uint64_t>(16ull << 20, g->chunk_bytes / 8);   // room for the carry in front of a chunk
    g->n_threads = n_threads > 0 ? n_threads : 8;
    g->t_open = std::chrono::steady_clock::now();
    g->stats.file_bytes = g->file_size;
    if (direct) g->fd_direct = open(path, O_RDONLY | O_DIRECT);   // refused by some filesystems (tmpfs): reads stay buffered
    if (numa) gpu_numa_cpus(device, &g->numa_node, g->numa_cpus);
    g->stats.direct_io = g->fd_direct >= 0 ? 1 : 0;
    g->stats.numa_node = g->numa_node;
    // compressed input?  gzip magic 1f 8b; BGZF if the first member carries the 'BC' extra subfield
    uint8_t magic[18] = {0};
    const bool gzip_magic = g->file_size >= 18 && pread(fd, magic, 18, 0) == 18 && magic[0] == 0x1f && magic[1] == 0x8b;
    const bool gz_on_device = gzip_magic && !bzq::bgzf_block_size(magic) && gpu_inflate;
    if (gz_on_device) g->gz_piece = std::max<uint64_t>(g->chunk_bytes / 2, std::min<uint64_t>(g->chunk_bytes, 64ull << 10));
    if (gz_on_device && getenv("BZQ_GZ_PIECE_MIB") && atoll(getenv("BZQ_GZ_PIECE_MIB")) > 0) g->gz_piece = std::min<uint64_t>(g->gz_piece.load(), (uint64_t)atoll(getenv("BZQ_GZ_PIECE_MIB")) << 20);   // (sweeps)   // (the first pieces; then by the file's compression ratio, up to a chunk: gz_fill_fifo)
    // (a .gz decoded on the device copies on the decoder's own streams: one stream less in the default class, whose four hardware queues
    // the decoder's find / copy / finder streams then have to themselves)
    bool ok = gz_on_device || bzq::cache::stream_pool().get(g->device, &g->copy_stream) == hipSuccess;
    if (ok) {   // the slots side by side: pinning a chunk-sized buffer takes ~30 ms, and three of them one after the other were most of an open
        bool slot_ok[bzq::INGEST_SLOTS];
        std::thread th[bzq::INGEST_SLOTS];
        for (int i = 0; i < bzq::INGEST_SLOTS; ++i)
            th[i] = std::thread([g, i, device, &slot_ok]() {
                slot_ok[i] = hipSetDevice(device) == hipSuccess && bzq::ingest_alloc_slot(g, i, g->chunk_bytes);   // (a .gz on the device reads its compressed pieces into the three slots' pinned buffers, two pieces ahead)
            });
        for (int i = 0; i < bzq::INGEST_SLOTS; ++i) { th[i].join(); ok = ok && slot_ok[i]; }
    }
    if (!ok) {
        err = std::string(who) + ": allocating the pinned / device chunk buffers failed";
        bzq::ingest_free(g);
        return BZQ_ERR_NOMEM;
    }
    if (gzip_magic) {
        if (bzq::bgzf_block_size(magic)) {
            g->compression = 2;
            if (gpu_inflate) {   // the compressed bytes travel, the device inflates (bzq_inflate.hpp)
                g->tab_cap = (int64_t)(g->chunk_bytes / 2048) + 4096;
                g->gpu_inflate = 1;
                g->inflate_ms = (gpu_inflate & 2) ? 1 : 0;
                if (const char* e = getenv("BZQ_INGEST_INFLATE_STREAMS")) g->n_inflate_streams = std::min(bzq::INGEST_SLOTS, std::max(1, atoi(e)));   // (measurements)
                bool gok = (g->bad_pinned = (unsigned long long*)bzq::cache::host_small().get(bzq::INGEST_SLOTS * sizeof(unsigned long long))) != nullptr &&
                           bzq::cache::get_device(g->device, bzq::INGEST_SLOTS * sizeof(unsigned long long), &g->bad_dev);
                for (int i = 0; i < bzq::INGEST_SLOTS && gok; ++i) { g->bad_pinned[i] = ~0ull; gok = bzq::ingest_alloc_inflate(g, i); }
                if (!gok) {
                    err = std::string(who) + ": allocating the buffers of the device inflate failed";
                    bzq::ingest_free(g);
                    return BZQ_ERR_NOMEM;
                }
            }
        } else if (gpu_inflate) {   // any other gzip file: rapidgzip's two-stage decode on the device (bzq_gzip.hpp)
            g->compression = 1;
            g->gz_cap = std::max<uint64_t>(6 * g->chunk_bytes, 64ull << 20);
            if (const char* e = getenv("BZQ_GZ_FIFO_KIB"))   // tests: a FIFO of a few chunks, so that small files walk through its wrap-around (never below 3 chunks)
                g->gz_cap = std::max<uint64_t>(3 * g->chunk_bytes, (uint64_t)atoll(e) << 10);
            int grc = bzq::gz::gz_open(device, &g->gz_dev, err);
            if (!grc) {
                // the pipeline of a FILE (round 5): the FIFO's copies are on the decoder's stream, so a piece's bytes are complete in stream
                // order and a call need not wait for its last kernels (defer_verify); those kernels then run beside the next piece's
                // decoders (chain_l2: the form that fits what the decoders leave of a CU) and the piece after next, on the device a
                // chain's length earlier, is searched behind its copy (early_find).  Steady state 15.8 -> 14.0 ms per 256 MiB piece
                // (profiles/r5_gzip_pipeline.md); the environment overrides each for A/B runs (gz_open)
                g->gz_dev->defer_verify = true;
                if (!getenv("BZQ_GZ_CHAIN_L2")) g->gz_dev->chain_l2 = true;
                if (!getenv("BZQ_GZ_EARLY_FIND")) g->gz_dev->early_find = true;
            }
            if (!grc && !bzq::cache::get_device(device, g->gz_cap + 64, &g->gz_fifo)) {
                err = std::string(who) + ": allocating the gzip FIFO failed"; grc = BZQ_ERR_NOMEM;
            }
            if (grc) { bzq::ingest_free(g); return grc; }
        } else {
            g->compression = 1;
            const int fd2 = dup(fd);
            g->gz = fd2 >= 0 ? gzdopen(fd2, "rb") : nullptr;
            if (!g->gz) {
                if (fd2 >= 0) close(fd2);
                err = std::string(who) + ": gzdopen failed for " + path;
                bzq::ingest_free(g);
                return BZQ_ERR_IO;
            }
            (void)gzbuffer(g->gz, 1u << 20);
        }
    }
    g->producer = std::thread(bzq::ingest_producer, g);
    *out = g;
    return 0;
}

// Where chunk k (carry + body) is assembled on the device: in front of the slot's body when the carry fits the reserve
// (no copy of the body), otherwise in big[k % INGEST_SLOTS], grown to fit (the caller then also copies the body there).  `quiesce`:
// the stream whose work may still touch an old big[] buffer.
static int ingest_place(bzq_ingest* g, int64_t k, uint64_t carry, hipStream_t quiesce, std::string& err, uint8_t** dst, bool* body_moves) {
    bzq::IngestSlot& s = g->slot[k % bzq::INGEST_SLOTS];
    if (carry <= g->reserve) { *dst = s.dev + (g->reserve - carry); *body_moves = false; return 0; }
    const uint64_t need = carry + s.len + 64;
    const int b = (int)(k % bzq::INGEST_SLOTS);
    if (g->big_cap[b] < need) {
        if (g->big[b]) { (void)hipStreamSynchronize(quiesce); (void)hipFree(g->big[b]); g->big[b] = nullptr; g->big_cap[b] = 0; }
        const uint64_t want = need + need / 4;
        if (hipMalloc((void**)&g->big[b], want) != hipSuccess) {
            (void)hipGetLastError();
            err = "ingest: cannot allocate " + std::to_string(want) + " bytes for a chunk whose carry (" + std::to_string(carry) + " bytes) exceeds the reserve";
            return BZQ_ERR_NOMEM;
        }
        g->big_cap[b] = want;
    }
    *dst = g->big[b];
    *body_moves = true;
    return 0;
}

// ---- file-chunk sharding: a rank's byte range of one file into device memory -------------------------------------------------
// (north_star "file-chunk sharding across the GPUs"; what FileReader.read_to_buffer does for the reference, io/readers.mojo:86-137,
// for the range [lo, hi) at once.)  Reader thread w takes pieces w, w + T, w + 2T ... of SHARD_PIECE bytes: pread() into one of its
// two pinned buffers, H2D on its own stream, the other buffer being read meanwhile -- no thread ever waits for another one, the
// copies of the T streams share the DMA engines.  The threads run on the CPUs of the GPU's NUMA node (option ingest_numa).
constexpr uint64_t SHARD_PIECE = 16ull << 20;

int32_t bzq_shard_read_range(bzq_ctx* c, const char* path, uint64_t lo, uint64_t hi, uint64_t halo_room, int32_t n_threads,
                             uint8_t** d_shard, uint64_t* n_out, uint64_t* capacity_out) {
    if (!c || !path || !d_shard || !n_out || !capacity_out || hi < lo) { if (c) c->err = "bzq_shard_read_range: bad arguments"; return BZQ_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { c->err = std::string("bzq_shard_read_range: cannot open ") + path; return BZQ_ERR_IO; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || hi > (uint64_t)st.st_size) {
        close(fd);
        c->err = std::string("bzq_shard_read_range: not a regular file, or the range ends behind it: ") + path;
        return BZQ_ERR_IO;
    }
    const uint64_t n = hi - lo;
    const uint64_t cap = ((n + std::max<uint64_t>(halo_room, 4ull << 20) + 64) + 15) & ~15ull;
    if (c->pending) HIPCHK(c, hipStreamSynchronize(c->stream));   // (a parse of the previous contents of the buffer may still run)
    int rc;
    if ((rc = ensure(c, c->shard_buf, cap))) { close(fd); return rc; }
    uint8_t* dst = (uint8_t*)c->shard_buf.p;
    const int64_t pieces = (int64_t)((n + SHARD_PIECE - 1) / SHARD_PIECE);
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 8, std::max<int64_t>(pieces, 1)));
    while ((int)c->shard_streams.size() < T) {
        hipStream_t q = nullptr;
        if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) { close(fd); c->err = "bzq_shard_read_range: hipStreamCreate"; return BZQ_ERR_HIP; }
        c->shard_streams.push_back(q);
    }
    while ((int)c->shard_events.size() < 2 * T) {
        hipEvent_t q = nullptr;
        if (hipEventCreateWithFlags(&q, hipEventDisableTiming) != hipSuccess) { close(fd); c->err = "bzq_shard_read_range: hipEventCreate"; return BZQ_ERR_HIP; }
        c->shard_events.push_back(q);
    }
    if ((int)c->shard_pin.size() < 2 * T) c->shard_pin.resize(2 * T, nullptr);
    std::vector<int> cpus;
    int node = -1;
    if (c->ingest_numa) gpu_numa_cpus(c->device, &node, cpus);
    std::atomic<int> fail{0};   // 1 = pinned allocation, 2 = read, 3 = HIP
    auto work = [&](int w) {
        if (hipSetDevice(c->device) != hipSuccess) { fail = 3; return; }
        if (!cpus.empty()) bzq::bind_to_cpus(cpus);
        hipStream_t q = c->shard_streams[w];
        bool used[2] = {false, false};
        int64_t i = 0;
        for (int64_t k = w; k < pieces && !fail; k += T, ++i) {
            const int b = (int)(i & 1);
            void*& pin = c->shard_pin[2 * w + b];
            if (!pin && bzq::cache::pinned_pool().get(c->device, SHARD_PIECE, &pin) != hipSuccess) { pin = nullptr; fail = 1; return; }   // (pinned here: T threads pin side by side)
            if (used[b] && hipEventSynchronize(c->shard_events[2 * w + b]) != hipSuccess) { fail = 3; return; }
            const uint64_t off = (uint64_t)k * SHARD_PIECE, len = std::min<uint64_t>(SHARD_PIECE, n - off);
            uint64_t got = 0;
            while (got < len) {
                const ssize_t r = pread(fd, (uint8_t*)pin + got, (size_t)(len - got), (off_t)(lo + off + got));
                if (r <= 0) { fail = 2; return; }
                got += (uint64_t)r;
            }
            bzq::cache::pinned_pool().pin(pin, 0, len);   // (lazy pinning: the pread above has just made the pages)
            if (hipMemcpyAsync(dst + off, pin, len, hipMemcpyHostToDevice, q) != hipSuccess || hipEventRecord(c->shard_events[2 * w + b], q) != hipSuccess) { fail = 3; return; }
            used[b] = true;
        }
        if (hipStreamSynchronize(q) != hipSuccess) fail = 3;
    };
    {
        std::vector<std::thread> th;
        for (int w = 1; w < T && pieces > 0; ++w) th.emplace_back(work, w);
        if (pieces > 0) work(0);
        for (auto& t : th) t.join();
    }
    close(fd);
    if (fail) {
        (void)hipGetLastError();
        c->err = fail == 1 ? "bzq_shard_read_range: pinning the staging buffers failed" : fail == 2 ? "bzq_shard_read_range: pread failed or the file was truncated"
                                                                                                    : "bzq_shard_read_range: a copy to the device failed";
        return fail == 1 ? BZQ_ERR_NOMEM : fail == 2 ? BZQ_ERR_IO : BZQ_ERR_HIP;
    }
    *d_shard = dst; *n_out = n; *capacity_out = cap;
    return 0;
}

int32_t bzq_ingest_open(bzq_ctx* c, const char* path, uint64_t chunk_bytes, int32_t n_threads, bzq_ingest** out) {
    if (!c || !path || !out) return BZQ_ERR_ARG;
    const int32_t rc = ingest_open_common(c->device, c->err, "bzq_ingest_open", path, chunk_bytes, n_threads, out, c->ingest_direct, c->ingest_numa,
                                          c->ingest_gpu_inflate ? (c->inflate_ms ? 3 : 1) : 0);
    if (rc == 0) {
        (*out)->ctx = c;
        // a new stream: the window follower of an earlier one must not judge this one's tail
        c->follow_on = false; c->stage_valid = false; c->records_before = -1;
        // While the reader threads fetch the first chunk: what the first parse would otherwise do before its first kernel -- the
        // arenas of a whole chunk, and the library's code object onto the device (the runtime loads it at the first launch:
        // ~20 ms of a fresh process, scripts/process_probe.sh).  Failures are the first submit's to report.
        if (c->n_submits == 0 && !c->cfg.views_only) {
            bzq_ingest* g = *out;
            uint64_t n0 = g->chunk_bytes + g->reserve;
            if (g->compression == 0) n0 = std::min<uint64_t>(n0, g->file_size);
            if (n0 && ensure_tile_arenas(c, n0) == 0 && ensure_col_arenas(c, n0) == 0)
                (void)ensure_record_arenas(c, (int64_t)(n0 / (uint64_t)std::max(4, c->cfg.min_record_bytes)) + 1024);
            hipLaunchKernelGGL(k_cumulate, dim3(1), dim3(BLOCK), 0, c->stream, (const int64_t*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr,
                               (int64_t)1, (int64_t)0, (int64_t*)nullptr, (int64_t*)nullptr);
            (void)hipGetLastError();
            c->err.clear();
        }
    }
    return rc;
}

// Parses the next chunk of the file.  `records_taken` = how many records of the PREVIOUS chunk the caller consumed
// (ignored on the first call); the rest, and the bytes behind them, are carried in front of this chunk.  Returns like
// bzq_chunk_result: 0 = records delivered and more input follows, > 0 = the stream's terminal FastxErrorCode (BZQ_EOF
// for a clean end; the chunk may still deliver records before it), < 0 runtime failure.  out->n_bytes / positions are
// relative to the chunk; *stream_pos (optional) receives the file offset of its first byte.
int32_t bzq_ingest_next(bzq_ingest* g, uint64_t records_taken, bzq_chunk* out, uint64_t* stream_pos) {
    if (!g || !out) return BZQ_ERR_ARG;
    bzq_ctx* c = g->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (g->finished) {
        memset(out, 0, sizeof(*out));
        out->status = g->final_status; out->error_record = -1;
        return g->final_status;
    }
    const int64_t k = g->next_k;
    // ---- carry of the previous chunk -----------------------------------------------------------------------
    uint64_t carry = 0, carry_src = 0;
    if (g->have_prev) {
        uint64_t cut = g->prev_res.bytes_consumed;
        if (records_taken < g->prev_res.n_records) {
            cut = 0;
            if (records_taken > 0) {
                int64_t e = 0;
                HIPCHK(c, hipMemcpy(&e, g->prev_res.d_record_end + (records_taken - 1), 8, hipMemcpyDeviceToHost));
                cut = (uint64_t)e + 1;
            }
        }
        carry = g->prev_n - cut;
        carry_src = cut;
        g->stats.records += std::min<uint64_t>(records_taken, g->prev_res.n_records);
    }
    // ---- wait for this chunk's H2D to be enqueued, then order the ctx stream behind it ---------------------------
    const auto tw = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return g->produced > k || g->stop; });
        if (g->produced <= k) { c->err = "bzq_ingest_next: " + (g->io_error.empty() ? std::string("reader stopped") : g->io_error); return BZQ_ERR_IO; }
    }
    bzq::IngestSlot& s = g->slot[k % bzq::INGEST_SLOTS];
    uint8_t* dst = nullptr;
    bool body_moves = false;
    {
        int prc;
        if ((prc = ingest_place(g, k, carry, c->stream, c->err, &dst, &body_moves))) return prc;
    }
    // the carry lands in FRONT of this chunk's body (the reserve), which the chunk's own copy / inflate does not touch: it
    // does not wait for them, and the previous chunk's buffer is released as soon as the carry is out of it -- so that the
    // producer can start chunk k+1 on the device while chunk k is still arriving (device inflate: two chunks' blocks in flight)
    if (carry) HIPCHK(c, hipMemcpyAsync(dst, g->prev_ptr + carry_src, carry, hipMemcpyDeviceToDevice, c->stream));
    if (g->have_prev) {
        HIPCHK(c, hipEventRecord(g->dev_free[(k - 1) % bzq::INGEST_SLOTS], c->stream));
        std::unique_lock<std::mutex> lk(g->mu);
        g->dev_free_valid[(k - 1) % bzq::INGEST_SLOTS] = true;
        g->released = k;
        g->cv.notify_all();
    }
    HIPCHK(c, hipStreamWaitEvent(c->stream, s.h2d_done, 0));
    if (body_moves && s.len) HIPCHK(c, hipMemcpyAsync(dst + carry, s.dev + g->reserve, s.len, hipMemcpyDeviceToDevice, c->stream));
    // the chunk starts wherever its carry starts: the kernels read the input through unaligned-typed vector loads
    const uint64_t n = carry + s.len;
    const uint64_t spos = s.file_off - carry;
    c->shard_mode = false;
    c->records_before = (int64_t)g->stats.records;   // records handed out so far: the tail log stays in stream order
    int rc = submit_common(c, dst, n, spos, s.eof ? 1 : 0, 0, 0, 0, 0, 10u, 0);
    if (rc < 0) return rc;
    rc = bzq_chunk_result(c, out);
    g->stats.wait_s += bzq::seconds_since(tw);
    if (rc < 0) return rc;
    if (g->gpu_inflate && g->bad_pinned[k % bzq::INGEST_SLOTS] != ~0ull) {   // (the stream is synchronised: the verdict travelled behind the kernel)
        c->err = "bzq_ingest_next: BGZF block " + std::to_string(g->bad_pinned[k % bzq::INGEST_SLOTS]) + " of the chunk at stream offset " + std::to_string(s.file_off) +
                 " failed to inflate (corrupt or truncated file)";
        return BZQ_ERR_IO;
    }
    if (stream_pos) *stream_pos = spos;
    g->stats.chunks += 1;
    g->stats.total_s = bzq::seconds_since(g->t_open);
    g->have_prev = true; g->prev_res = *out; g->prev_n = n; g->prev_ptr = dst; g->prev_stream_pos = spos;
    g->next_k = k + 1;
    if (out->status != BZQ_OK) {   // terminal: EOF or the first failing record
        g->finished = true; g->final_status = out->status == BZQ_EOF ? BZQ_EOF : out->status;
        g->stats.records += out->n_records;
        std::unique_lock<std::mutex> lk(g->mu);
        g->stop = true;
        g->cv.notify_all();
    } else if (s.eof) {
        g->finished = true; g->final_status = BZQ_EOF;
    }
    return rc;
}

int32_t bzq_ingest_get_stats(const bzq_ingest* g, bzq_ingest_stats* out) {
    if (!g || !out) return BZQ_ERR_ARG;
    *out = g->stats;
    return 0;
}

void bzq_ingest_close(bzq_ingest* g) {
    if (g && g->ctx) {
        g->ctx->follow_on = false; g->ctx->stage_valid = false; g->ctx->records_before = -1;   // (the stream is over: see bzq_ingest_open)
        // who may still touch the chunk buffers: the parser's stream, and in views mode the consumer stream (views point into the chunk)
        g->quiesce_device = false;
        g->quiesce_stream = g->ctx->stream;
        g->quiesce_stream2 = g->ctx->cfg.views_only ? g->ctx->consumer_stream : nullptr;
    }
    bzq::ingest_free(g);
}

// ---- the same pipeline in front of the FASTA parser (bzq_fasta.hip) -----------------------------------------------------

// internal accessors of the other translation unit
int32_t bzq_fasta_device_(const bzq_fasta* h);
void bzq_fasta_set_error_(bzq_fasta* h, const char* msg);

struct bzq_fasta_ingest {
    bzq_ingest* g = nullptr;
    bzq_fasta* h = nullptr;
    hipStream_t aux = nullptr;          // orders the carry copy behind the chunk's H2D
    uint64_t prev_consumed = 0, line_base = 0, record_base = 0;
    std::string err;
};

int32_t bzq_fasta_ingest_open(bzq_fasta* h, const char* path, uint64_t chunk_bytes, int32_t n_threads, bzq_fasta_ingest** out) {
    if (!h || !path || !out) return BZQ_ERR_ARG;
    *out = nullptr;
    bzq_fasta_ingest* f = new bzq_fasta_ingest();
    f->h = h;
    const int32_t rc = ingest_open_common(bzq_fasta_device_(h), f->err, "bzq_fasta_ingest_open", path, chunk_bytes, n_threads, &f->g);
    if (rc != 0 || hipStreamCreateWithFlags(&f->aux, hipStreamNonBlocking) != hipSuccess) {
        bzq_fasta_set_error_(h, rc ? f->err.c_str() : "bzq_fasta_ingest_open: stream creation failed");
        if (f->g) bzq::ingest_free(f->g);
        delete f;
        return rc ? rc : BZQ_ERR_HIP;
    }
    *out = f;
    return 0;
}

void bzq_fasta_ingest_close(bzq_fasta_ingest* f) {
    if (!f) return;
    if (f->g) { (void)hipSetDevice(f->g->device); bzq::ingest_free(f->g); }
    if (f->aux) (void)hipStreamDestroy(f->aux);
    delete f;
}

int32_t bzq_fasta_ingest_get_stats(const bzq_fasta_ingest* f, bzq_ingest_stats* out) {
    if (!f || !out) return BZQ_ERR_ARG;
    *out = f->g->stats;
    return 0;
}

// The next chunk of the file through bzq_fasta_parse (same result struct, offsets relative to the chunk; *stream_pos = file
// offset of its first byte).  Every record a chunk delivers is taken; the bytes of the open record are carried in front
// of the next chunk on the device.  BZQ_FASTA_NEED_MORE never comes out: the next piece of the file is appended instead.
int32_t bzq_fasta_ingest_next(bzq_fasta_ingest* f, bzq_fasta_chunk* out, uint64_t* stream_pos) {
    if (!f || !out) return BZQ_ERR_ARG;
    bzq_ingest* g = f->g;
    auto fail = [&](const std::string& m, int32_t code) { bzq_fasta_set_error_(f->h, m.c_str()); return code; };
    if (hipSetDevice(g->device) != hipSuccess) return fail("bzq_fasta_ingest_next: hipSetDevice failed", BZQ_ERR_HIP);
    if (g->finished) {
        memset(out, 0, sizeof(*out));
        out->status = g->final_status;
        return 0;
    }
    for (;;) {
        const int64_t k = g->next_k;
        uint64_t carry = 0, carry_src = 0;
        if (g->have_prev) {
            carry = g->prev_n - f->prev_consumed;
            carry_src = f->prev_consumed;
        }
        const auto tw = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->produced > k || g->stop; });
            if (g->produced <= k) return fail("bzq_fasta_ingest_next: " + (g->io_error.empty() ? std::string("reader stopped") : g->io_error), BZQ_ERR_IO);
        }
        bzq::IngestSlot& s = g->slot[k % bzq::INGEST_SLOTS];
        bool ok = true;
        uint8_t* dst = nullptr;
        bool body_moves = false;
        {   // a record longer than the reserve (a chromosome): the chunk is assembled in a buffer grown to fit
            std::string perr;
            const int prc = ingest_place(g, k, carry, f->aux, perr, &dst, &body_moves);
            if (prc) return fail("bzq_fasta_ingest_next: " + perr, prc);
        }
        // (the carry goes in front of the body and does not wait for the chunk's own copy / inflate: bzq_ingest_next)
        if (carry) ok = hipMemcpyAsync(dst, g->prev_ptr + carry_src, carry, hipMemcpyDeviceToDevice, f->aux) == hipSuccess;
        if (ok && g->have_prev) {   // the previous chunk's device buffer may now be refilled (behind the carry copy)
            ok = hipEventRecord(g->dev_free[(k - 1) % bzq::INGEST_SLOTS], f->aux) == hipSuccess;
            std::unique_lock<std::mutex> lk(g->mu);
            g->dev_free_valid[(k - 1) % bzq::INGEST_SLOTS] = true;
            g->released = k;
            g->cv.notify_all();
        }
        if (ok) ok = hipStreamWaitEvent(f->aux, s.h2d_done, 0) == hipSuccess;
        if (ok && body_moves && s.len) ok = hipMemcpyAsync(dst + carry, s.dev + g->reserve, s.len, hipMemcpyDeviceToDevice, f->aux) == hipSuccess;
        if (!ok || hipStreamSynchronize(f->aux) != hipSuccess) return fail("bzq_fasta_ingest_next: a HIP call failed", BZQ_ERR_HIP);
        if (g->gpu_inflate && g->bad_pinned[k % bzq::INGEST_SLOTS] != ~0ull)
            return fail("bzq_fasta_ingest_next: BGZF block " + std::to_string(g->bad_pinned[k % bzq::INGEST_SLOTS]) + " of the chunk at stream offset " +
                        std::to_string(s.file_off) + " failed to inflate (corrupt or truncated file)", BZQ_ERR_IO);
        const uint64_t n = carry + s.len, spos = s.file_off - carry;
        const int32_t rc = bzq_fasta_parse(f->h, dst, n, s.eof ? 1 : 0, spos, f->line_base, f->record_base, out);
        g->stats.wait_s += bzq::seconds_since(tw);
        if (rc < 0) return rc;
        if (stream_pos) *stream_pos = spos;
        g->stats.chunks += 1;
        g->stats.total_s = bzq::seconds_since(g->t_open);
        g->have_prev = true; g->prev_n = n; g->prev_ptr = dst; g->prev_stream_pos = spos;
        g->next_k = k + 1;
        g->stats.records += (uint64_t)out->n_records;
        f->record_base += (uint64_t)out->n_records;
        if (out->status == BZQ_OK || out->status == BZQ_FASTA_NEED_MORE) {
            f->prev_consumed = out->bytes_consumed;
            f->line_base += (uint64_t)out->lines_consumed;
            if (out->status == BZQ_OK) return 0;
            continue;   // no record closed inside this chunk: take the next piece of the file behind it
        }
        g->finished = true; g->final_status = out->status;
        std::unique_lock<std::mutex> lk(g->mu);
        g->stop = true;
        g->cv.notify_all();
        return 0;
    }
}

// ---- upload of a host FastqBatch (record_batch.mojo:308-411) ---------------------------------------------------

// The reference's upload_batch_to_device: 5 pinned allocations, 5 memcpys, 5 device allocations, 5 H2D copies and 3
// synchronize() per batch.  Here: ONE device allocation holding the five arrays (each 16-byte aligned), five async
// copies on the ctx stream, one synchronize.  Only needed for a batch that outlived its chunk (while the chunk is
// live, bzq_batch_view is zero-copy); the caller owns the result and frees it with bzq_release_batch.
int32_t bzq_upload_batch(bzq_ctx* c, const bzq_host_batch* h, bzq_device_batch* out) {
    if (!c || !h || !out || h->num_records < 0) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    memset(out, 0, sizeof(*out));
    out->quality_offset = h->quality_offset;
    const int64_t n = h->num_records;
    if (n == 0) return 0;
    if (!h->ends || !h->id_ends) return BZQ_ERR_ARG;
    const int64_t seq_len = h->ends[n - 1], id_len = h->id_ends[n - 1];   // inclusive running sums (Q11)
    if (seq_len < 0 || id_len < 0 || (seq_len && (!h->quality_bytes || !h->sequence_bytes)) || (id_len && !h->id_bytes)) return BZQ_ERR_ARG;
    auto up16 = [](int64_t v) { return (size_t)((v + 15) & ~(int64_t)15); };
    const size_t o_q = 0, o_s = o_q + up16(seq_len), o_i = o_s + up16(seq_len), o_e = o_i + up16(id_len), o_ie = o_e + up16(n * 8),
                 total = o_ie + up16(n * 8);
    uint8_t* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) { c->err = std::string("bzq_upload_batch: hipMalloc: ") + hipGetErrorString(e); return BZQ_ERR_NOMEM; }
    bool ok = true;
    auto cp = [&](size_t off, const void* src, size_t bytes) {
        if (bytes && hipMemcpyAsync(d + off, src, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) ok = false;
    };
    cp(o_q, h->quality_bytes, (size_t)seq_len); cp(o_s, h->sequence_bytes, (size_t)seq_len); cp(o_i, h->id_bytes, (size_t)id_len);
    cp(o_e, h->ends, (size_t)n * 8); cp(o_ie, h->id_ends, (size_t)n * 8);
    if (!ok || hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(d);
        c->err = "bzq_upload_batch: host to device copy failed";
        return BZQ_ERR_HIP;
    }
    out->num_records = n; out->seq_len = seq_len; out->total_id_bytes = id_len; out->sequence_bytes = seq_len;
    out->qual_buffer = d + o_q; out->sequence_buffer = d + o_s; out->id_buffer = d + o_i;
    out->ends = (const int64_t*)(d + o_e); out->id_ends = (const int64_t*)(d + o_ie);
    out->first_record = ~0ull;   // not a view of the chunk
    return 0;
}

int32_t bzq_release_batch(bzq_ctx* c, bzq_device_batch* b) {
    if (!c || !b) return BZQ_ERR_ARG;
    if (b->first_record != ~0ull) { c->err = "bzq_release_batch: not an uploaded batch (views of the chunk are owned by the ctx)"; return BZQ_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    if (b->qual_buffer) HIPCHK(c, hipFree((void*)b->qual_buffer));
    memset(b, 0, sizeof(*b));
    return 0;
}

// ---- device-side consumers of a DeviceFastqBatch (bzq_consumers.hpp) --------------------------------------------

// the nw_gpu example's scores of a batch: one thread per record for references of at most 64 bases, one wave per record beyond
static void launch_nw(hipStream_t cs, const uint8_t* d_ref, int ref_len, const bzq_device_batch* b, int32_t* d_scores) {
    const int64_t n = b->num_records;
    const dim3 tg((unsigned)((n + BLOCK - 1) / BLOCK));
    // (valid bytes of the sequence column: a batch struct filled by hand may lack sequence_bytes; the records then end at seq_len)
    const int64_t col_len = b->sequence_bytes > 0 ? b->sequence_bytes : b->seq_len;
#define BZQ_NW_T(RL) hipLaunchKernelGGL(k_nw_scores_t<RL>, tg, dim3(BLOCK), 0, cs, d_ref, ref_len, b->sequence_buffer, b->ends, n, col_len, d_scores)
    if (ref_len <= 16) BZQ_NW_T(16);
    else if (ref_len <= 32) BZQ_NW_T(32);
    else if (ref_len <= 40) BZQ_NW_T(40);
    else if (ref_len <= 48) BZQ_NW_T(48);
    else if (ref_len <= 64) BZQ_NW_T(64);
    else hipLaunchKernelGGL(k_nw_scores, dim3((unsigned)((n + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0, cs, d_ref, ref_len, b->sequence_buffer, b->ends, n, d_scores);
#undef BZQ_NW_T
}

int32_t bzq_batch_nw_scores(bzq_ctx* c, const bzq_device_batch* b, const uint8_t* ref, int32_t ref_len, int32_t* d_scores) {
    if (!c || !b || ref_len < 0 || (ref_len && !ref) || (b->num_records && !d_scores)) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = c->consumer_stream ? c->consumer_stream : c->stream;
    if (b->num_records <= 0) return 0;
    int rc;
    if ((rc = ensure(c, c->consumer_scratch, 4096))) return rc;
    const int copy = ref_len > NW_MAX_LEN ? NW_MAX_LEN : ref_len;   // longer references score 0 like the example
    if (copy) HIPCHK(c, hipMemcpyAsync(c->consumer_scratch.p, ref, (size_t)copy, hipMemcpyHostToDevice, cs));
    launch_nw(cs, (const uint8_t*)c->consumer_scratch.p, (int)ref_len, b, d_scores);
    HIPCHK(c, hipStreamSynchronize(cs));   // the host reference bytes may go away after the call
    return 0;
}

int32_t bzq_batch_quality_sums(bzq_ctx* c, const bzq_device_batch* b, int64_t* d_sums) {
    if (!c || !b || (b->num_records && !d_sums)) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = c->consumer_stream ? c->consumer_stream : c->stream;
    const int64_t n = b->num_records;
    if (n <= 0) return 0;
    // short reads: a workgroup per 256 records (coalesced span + LDS accumulators); long reads: a wave per record
    if (b->seq_len / n < 1024)
        hipLaunchKernelGGL(k_quality_sums_block<false>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, cs,
                           b->qual_buffer, b->ends, n, b->seq_len, (int)b->quality_offset, d_sums);
    else
        hipLaunchKernelGGL(k_quality_sums<false>, dim3((unsigned)((n + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0, cs,
                           b->qual_buffer, b->ends, n, b->seq_len, (int)b->quality_offset, d_sums);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("k_quality_sums: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    return 0;
}

int32_t bzq_column_gc_counts(bzq_ctx* c, const uint8_t* d_col, const int64_t* d_ends, int64_t n, int64_t col_len, int64_t* d_counts) {
    if (!c || n < 0 || (n && (!d_col || !d_ends || !d_counts))) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = c->consumer_stream ? c->consumer_stream : c->stream;
    if (n == 0) return 0;
    if (col_len / n < 1024)
        hipLaunchKernelGGL(k_quality_sums_block<true>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, cs, d_col, d_ends, n,
                           col_len, 0, d_counts);
    else
        hipLaunchKernelGGL(k_quality_sums<true>, dim3((unsigned)((n + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0, cs, d_col, d_ends,
                           n, col_len, 0, d_counts);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("k_quality_sums<GC>: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    HIPCHK(c, hipStreamSynchronize(cs));
    return 0;
}

int32_t bzq_batch_quality_by_position(bzq_ctx* c, const bzq_device_batch* b, int32_t max_positions, uint64_t* counts) {
    if (!c || !b || max_positions <= 0 || !counts) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = c->consumer_stream ? c->consumer_stream : c->stream;
    const size_t bytes = (size_t)max_positions * QP_BINS * 8;
    int rc;
    if ((rc = ensure(c, c->qpos_scratch, bytes))) return rc;
    HIPCHK(c, hipMemsetAsync(c->qpos_scratch.p, 0, bytes, cs));
    if (b->num_records > 0) {
        const dim3 grid((unsigned)((max_positions + QP_POS - 1) / QP_POS), (unsigned)((b->num_records + QP_RECS - 1) / QP_RECS));
        hipLaunchKernelGGL(k_quality_by_position, grid, dim3(BLOCK), 0, cs, b->qual_buffer, b->ends, b->num_records, (int)max_positions,
                           (u64*)c->qpos_scratch.p);
    }
    HIPCHK(c, hipMemcpyAsync(counts, c->qpos_scratch.p, bytes, hipMemcpyDeviceToHost, cs));
    HIPCHK(c, hipStreamSynchronize(cs));
    return 0;
}

int32_t bzq_batch_nw_scores_dev(bzq_ctx* c, const bzq_device_batch* b, const uint8_t* d_ref, int32_t ref_len, int32_t* d_scores) {
    if (!c || !b || ref_len < 0 || (ref_len && !d_ref) || (b->num_records && !d_scores)) return BZQ_ERR_ARG;
    if (ref_len > NW_MAX_LEN) { c->err = "bzq_batch_nw_scores_dev: ref_len > 256 (the synchronous call scores such a reference 0 like the example; here it is refused)"; return BZQ_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = c->consumer_stream ? c->consumer_stream : c->stream;
    const int64_t n = b->num_records;
    if (n <= 0) return 0;
    launch_nw(cs, d_ref, (int)ref_len, b, d_scores);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("k_nw_scores: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    return 0;
}

int32_t bzq_batch_quality_by_position_acc(bzq_ctx* c, const bzq_device_batch* b, int32_t max_positions, uint64_t* d_counts) {
    if (!c || !b || max_positions <= 0 || !d_counts) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = c->consumer_stream ? c->consumer_stream : c->stream;
    if (b->num_records <= 0) return 0;
    const dim3 grid((unsigned)((max_positions + QP_POS - 1) / QP_POS), (unsigned)((b->num_records + QP_RECS - 1) / QP_RECS));
    hipLaunchKernelGGL(k_quality_by_position, grid, dim3(BLOCK), 0, cs, b->qual_buffer, b->ends, b->num_records, (int)max_positions, (u64*)d_counts);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->err = std::string("k_quality_by_position: ") + hipGetErrorString(le); return BZQ_ERR_HIP; }
    return 0;
}

int32_t bzq_consumer_synchronize(bzq_ctx* c) {
    if (!c) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->consumer_stream ? c->consumer_stream : c->stream));
    return 0;
}

int32_t bzq_column_histogram(bzq_ctx* c, const uint8_t* d_col, uint64_t n, uint64_t* hist) {
    if (!c || !hist || (n && !d_col)) return BZQ_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = c->consumer_stream ? c->consumer_stream : c->stream;
    int rc;
    if ((rc = ensure(c, c->consumer_scratch, 4096))) return rc;
    u64* d_h = (u64*)c->consumer_scratch.p + 64;   // behind the reference bytes
    HIPCHK(c, hipMemsetAsync(d_h, 0, 256 * 8, cs));
    if (n) {
        const uint64_t steps = (n + (uint64_t)BLOCK * 16 - 1) / ((uint64_t)BLOCK * 16);
        const unsigned grid = (unsigned)std::min<uint64_t>(steps, (uint64_t)c->num_cu * 8);
        hipLaunchKernelGGL(k_byte_histogram, dim3(grid), dim3(BLOCK), 0, cs, d_col, (int64_t)n, d_h);
    }
    HIPCHK(c, hipMemcpyAsync(hist, d_h, 256 * 8, hipMemcpyDeviceToHost, cs));
    HIPCHK(c, hipStreamSynchronize(cs));
    return 0;
}

#include "bzq_fasta_shard.hpp"

} // extern "C"
