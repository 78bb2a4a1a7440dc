/*
 * blazeseq_hip.h -- C ABI of libblazeseq_hip.so: the MI355X (gfx950) FASTQ batch-parse path.
 *
 * This is the drop-in boundary for BlazeSeq's hot path.  Every entry point names the reference
 * interface it replaces (file:line relative to the BlazeSeq tree).  Plain pointers and sizes only:
 * a Mojo host binds these with OwnedDLHandle/get_function exactly like the reference binds libz
 * (blazeseq/io/readers.mojo:226-280); INTEGRATION.md shows that shim.
 *
 * Conventions
 *   - every call returns int32: 0 = OK, < 0 = runtime failure (HIP / allocation / misuse; text via
 *     bzq_last_error), > 0 = a FastxErrorCode (blazeseq/errors.mojo:33-68) where documented.
 *   - one bzq_ctx per host thread (the reference parser is single-threaded, README.md:113); a ctx
 *     owns its HIP stream, its device arenas and the result of the most recent chunk.
 *   - device pointers returned in bzq_chunk / bzq_device_batch / bzq_device_views are owned by the ctx and
 *     stay valid until the SECOND following bzq_submit_* / bzq_ingest_next on that ctx: the ctx keeps two
 *     sets of output arenas and alternates between them, so a consumer kernel on another stream may
 *     still read chunk k while chunk k+1 is parsed (the device analogue of "FastqView is valid until
 *     the next parser call", blazeseq/fastq/record.mojo:437-440, one chunk deeper).  The INPUT bytes a
 *     bzq_device_views points into belong to the caller (bzq_submit_chunk_device) or are valid until the
 *     next submit (bzq_submit_chunk_host).  Option "double_buffer" = 0 keeps one set (half the memory,
 *     results valid until the next submit).
 *   - all positions are int64 byte offsets relative to the first byte of the submitted chunk.
 */
#ifndef BLAZESEQ_HIP_H
#define BLAZESEQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the declarations of this header are exported (BZQ_BUILDING is set
 * by the library's own Makefile only; a client never sees the pragma). */
#if defined(BZQ_BUILDING) && defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define BZQ_ABI_VERSION 2   /* 2: bzq_chunk.d_ends / d_id_ends are produced on demand (bzq_chunk_cumulative_ends) */

/* FastxErrorCode, blazeseq/errors.mojo:33-68 (values are part of the ABI). */
enum {
    BZQ_OK = 0,
    BZQ_ID_NO_AT = 1,
    BZQ_SEP_NO_PLUS = 2,
    BZQ_SEQ_QUAL_LEN_MISMATCH = 3,
    BZQ_ASCII_INVALID = 4,
    BZQ_QUALITY_OUT_OF_RANGE = 5,
    BZQ_EOF = 6,
    BZQ_UNEXPECTED_EOF = 7,
    BZQ_BUFFER_EXCEEDED = 8,
    BZQ_BUFFER_AT_MAX = 9,
    BZQ_OTHER = 10
};

/* runtime failures (negative) */
enum {
    BZQ_ERR_HIP = -1,       /* a HIP call failed */
    BZQ_ERR_ARG = -2,       /* bad argument / call order */
    BZQ_ERR_NOMEM = -3,     /* arena too small for the chunk and growth refused */
    BZQ_ERR_NO_DEVICE = -4, /* no gfx950 device visible: the product path has no CPU fallback */
    BZQ_ERR_IO = -5
};

/* ParserConfig (blazeseq/fastq/parser.mojo:33-74) + QualitySchema triple
 * (blazeseq/fastq/quality_schema.mojo:9-31) + the parser's batch size (parser.mojo:125-145). */
typedef struct bzq_config {
    int64_t buffer_capacity;        /* ParserConfig.buffer_capacity (default 256 KiB, CONSTS.mojo:26).
                                       The GPU path has no such buffer; the value only decides which
                                       records the reference would have refused (BUFFER_EXCEEDED) */
    int64_t buffer_max_capacity;    /* default 2^30, CONSTS.mojo:28 */
    int32_t buffer_growth_enabled;  /* default 0 */
    int32_t check_ascii;            /* default 0 */
    int32_t check_quality;          /* default 0 */
    uint8_t q_lower, q_upper, q_offset, _pad0; /* QualitySchema.LOWER/UPPER/OFFSET */
    int32_t batch_size;             /* DEFAULT_BATCH_SIZE = 4096, CONSTS.mojo:31 */
    int32_t compat_simd_width;      /* 0 (default): quality bounds inclusive [LOWER,UPPER] (the scalar
                                       branch, record.mojo:99-102).  16/32/64: also reproduce the SIMD
                                       body's `>=` for the first floor(n/W)*W quality bytes
                                       (record.mojo:90-97) */
    int32_t emit_offsets;           /* also materialise RecordOffsets columns (utils.mojo:37-93) */
    int32_t views_only;             /* views() mode (parser.mojo:253-258): no columns, every record as RecordOffsets +
                                       its stripped-id span into the chunk (zero copy); see bzq_views() */
    int64_t max_chunk_bytes;        /* device arena sizing; chunks larger than this re-size the arena */
    int64_t pass_bytes;             /* bytes handled per kernel round (0 = library default) */
    int32_t min_record_bytes;       /* sizing hint for the per-record arrays (default 32); an input
                                       with shorter records transparently re-sizes and re-runs */
    int32_t _pad2;
} bzq_config;

typedef struct bzq_ctx bzq_ctx;

/* One parsed chunk.  Mirrors what FastqParser.next_view/next_batch expose record by record
 * (parser.mojo:159-170, 239-251), for all records of the chunk at once. */
typedef struct bzq_chunk {
    uint64_t n_bytes;          /* bytes submitted */
    uint64_t n_records;        /* records delivered: complete, and before the first failing record */
    uint64_t bytes_consumed;   /* chunk offset after the last delivered record; the caller carries
                                  [bytes_consumed, n_bytes) into its next chunk (the chunk-level
                                  analogue of SearchPhase resume, utils.mojo:485-487) */
    uint64_t total_newlines;
    int32_t status;            /* BZQ_OK: more input may follow (is_eof was 0);
                                  BZQ_EOF: clean end of stream; otherwise the FastxErrorCode the
                                  reference parser raises after delivering n_records records */
    int32_t tail_phase;        /* SearchPhase of the undelivered tail (0..3) */
    int64_t error_record;      /* chunk-local index of the failing record, -1 if none */
    uint64_t seq_bytes, qual_bytes, id_bytes; /* column lengths over the delivered records */
    /* FastqBatch columns for the whole chunk (record_batch.mojo:19-41), device resident */
    const uint8_t* d_seq;      /* u8[seq_bytes]  concatenated sequence lines */
    const uint8_t* d_qual;     /* u8[qual_bytes] concatenated quality lines */
    const uint8_t* d_id;       /* u8[id_bytes]   concatenated stripped ids (no '@') */
    /* CHUNK-cumulative running sums (i64[n_records]; Q11): an artefact of parsing a chunk at a time -- the reference has only the
     * per-batch arrays below, and nothing on the batches() path reads these.  NULL until bzq_chunk_cumulative_ends() is called
     * for the chunk (ABI 2), unless the chunk went through the older per-record pass (batch_size < 256, pass_bytes, the
     * compat_simd_width quirk, option "cumulative_ends" = 1), which fills them with every chunk. */
    const int64_t* d_ends;     /* inclusive running sum of quality lengths */
    const int64_t* d_id_ends;  /* inclusive running sum of id lengths */
    const int64_t* d_batch_ends;    /* same, restarted every batch_size records: exactly the
                                       `_ends` of the FastqBatch that next_batch would build */
    const int64_t* d_batch_id_ends;
    const int64_t* d_record_end;    /* i64[n_records] offset of the record's terminating '\n'
                                       (RecordOffsets.record_end, utils.mojo:61) */
    /* RecordOffsets columns (absolute chunk offsets), NULL unless config.emit_offsets */
    const int64_t* d_header_start;
    const int64_t* d_seq_start;
    const int64_t* d_sep_start;
    const int64_t* d_qual_start;
    /* timing of the kernels of this chunk, hipEvent on the ctx stream (milliseconds).  Since round 5 (lean submit): ms_total is ONE
     * interval, the submit's first event to its last (rounds 1-4: the sum of two intervals around a copy); the four parts are filled
     * only with option "timing_detail" = 1 (three more events per submit), else 0; ms_rebase = what follows the last emit (k_tail /
     * k_finish, or k_rebase where the per-record pass still runs). */
    float ms_total;            /* first kernel start -> last kernel end */
    float ms_aggregate, ms_scan, ms_emit, ms_rebase;
    uint32_t n_passes;
    uint32_t chunk_serial;     /* which submit of the ctx this chunk was (1, 2, ...; the low 32 bits): bzq_chunk_cumulative_ends refuses a
                                  bzq_chunk whose output set has been written again since (same pointers, another chunk).  Occupies what
                                  was padding in ABI 2 (same layout); the dense-tile diagnostic that sat here is the query "dense_tiles" */
    /* views mode only (config.views_only): the stripped id of record r is chunk[d_id_start[r] .. + d_id_len[r])
     * (FastqView.id(), record.mojo:431-472); the three column pointers and d_*ends are NULL in that mode */
    const int64_t* d_id_start;
    const int32_t* d_id_len;
} bzq_chunk;

/* Views of records [first_record, first_record + num_records) of the current chunk (views mode): FastqView
 * (record.mojo:431-550) for many records at once -- spans into the chunk instead of copies.  Record r (relative to
 * first_record): id = chunk[id_start[r] .. +id_len[r]); sequence = chunk[seq_start[r] .. sep_start[r] - 1);
 * quality = chunk[qual_start[r] .. record_end[r]).  The offset arrays follow the two-chunk lifetime rule above. */
typedef struct bzq_device_views {
    int64_t num_records;
    const uint8_t* chunk;          /* the submitted chunk on the device */
    const int64_t* header_start;   /* RecordOffsets columns (utils.mojo:37-93), absolute chunk offsets */
    const int64_t* seq_start;
    const int64_t* sep_start;
    const int64_t* qual_start;
    const int64_t* record_end;
    const int64_t* id_start;
    const int32_t* id_len;
    uint64_t first_record;
} bzq_device_views;

/* DeviceFastqBatch (blazeseq/fastq/record_batch.mojo:210-220): what FastqBatch.to_device(ctx)
 * returns (record_batch.mojo:89-90, 404-411).  Zero-copy slices of the chunk columns. */
typedef struct bzq_device_batch {
    int64_t num_records;
    int64_t seq_len;           /* ends[num_records-1] */
    int64_t total_id_bytes;    /* id_ends[num_records-1] */
    uint8_t quality_offset;    /* always 33 on the parser path (parser.mojo:243) */
    uint8_t _pad[7];
    const uint8_t* qual_buffer;
    const uint8_t* sequence_buffer;
    const int64_t* ends;       /* inclusive, relative to this batch */
    const uint8_t* id_buffer;
    const int64_t* id_ends;
    uint64_t first_record;     /* chunk-local index of record 0 of this batch */
    int64_t sequence_bytes;    /* valid bytes in sequence_buffer.  == seq_len, except for the batch that holds an
                                  unterminated last record whose sequence and quality lengths differ (no structure
                                  check for it, parser.mojo:464-475): the reference's FastqBatch._sequence_bytes then
                                  holds the true sequence bytes while _ends / seq_len follow the quality lengths */
} bzq_device_batch;

/* FastqBatch on the host (record_batch.mojo:19-41), filled by bzq_batch_to_host: the caller
 * provides buffers of at least the sizes in the matching bzq_device_batch. */
typedef struct bzq_host_batch {
    int64_t num_records;
    uint8_t* quality_bytes;    /* seq_len bytes */
    uint8_t* sequence_bytes;   /* bzq_device_batch.sequence_bytes bytes (== seq_len but for the case described there) */
    uint8_t* id_bytes;         /* total_id_bytes bytes */
    int64_t* ends;             /* num_records */
    int64_t* id_ends;          /* num_records */
    uint8_t quality_offset;
    uint8_t _pad[7];
} bzq_host_batch;

/* ---- configuration -------------------------------------------------------------------------- */

int32_t bzq_abi_version(void);
/* ParserConfig() defaults, parser.mojo:60-74; schema = generic */
void bzq_config_default(bzq_config* cfg);
/* _parse_schema, blazeseq/utils.mojo:612-637.  Returns 1 for a known name, 0 when it fell back to
 * generic (the reference prints a warning and continues). */
int32_t bzq_schema_from_name(const char* name, uint8_t* lower, uint8_t* upper, uint8_t* offset);
/* What `simd_width_of[DType.uint8]()` is on THIS host (64 with AVX-512BW, 32 with AVX2, else 16): the W of the reference's
 * SIMD quality check (record.mojo:76-104).  A host that wants to be bit-exact with the reference BINARY running on the same
 * machine passes it as bzq_config.compat_simd_width; the library default 0 keeps the documented inclusive bounds. */
int32_t bzq_host_simd_width(void);
/* FastxErrorCode.message(), blazeseq/errors.mojo:71-90 */
const char* bzq_message_for_code(int32_t code);

/* ---- context -------------------------------------------------------------------------------- */

/* FastqParser.__init__ (parser.mojo:89-145): one per parser.  `device` is the HIP ordinal.
 * Fails with BZQ_ERR_NO_DEVICE when no GPU is present -- there is no CPU fallback. */
int32_t bzq_create(int32_t device, const bzq_config* cfg, bzq_ctx** out);
void bzq_destroy(bzq_ctx* ctx);
const char* bzq_last_error(const bzq_ctx* ctx);   /* ctx may be NULL: last create() failure */
/* Use a caller-owned hipStream_t (e.g. torch's current stream) instead of the ctx's own. */
int32_t bzq_set_stream(bzq_ctx* ctx, void* hip_stream);
/* The stream the bzq_batch_* / bzq_column_* consumer kernels are launched on (NULL = the ctx stream).  With a stream
 * of its own a consumer of chunk k overlaps the parse of chunk k+1 (the results of chunk k stay valid, see above).
 * Priority classes: the parser's own stream is of the HIGHEST class, the library's helper streams (copies, inflate, block finder) of the
 * LOWEST (the runtime maps streams onto a few hardware queues per class, and a stream behind a 5 ms inflate kernel in a shared queue
 * waits for it).  Which queue a caller's stream gets depends on how many streams the process created before it; measured (bench.py
 * pipeline_mode, round 6): a consumer stream of the HIGHEST class kept its rate in every context, one of the default class lost 13 %
 * in a process that had created many streams. */
int32_t bzq_set_consumer_stream(bzq_ctx* ctx, void* hip_stream);
int32_t bzq_get_config(const bzq_ctx* ctx, bzq_config* out);
/* Run-time knobs that are not part of ParserConfig: "pass_bytes" (bytes per kernel round),
 * "timing_detail" (per-kernel hipEvent timing in bzq_chunk), "force_dense" (tests: route every
 * tile through the serial in-kernel path), "records_before" (a stream submitted in SEVERAL chunks: the
 * number of records delivered from earlier chunks, set before each bzq_submit_chunk_*, 0 for the stream's first
 * chunk; the ctx then follows the reference's BufferedReader window over the records handed out -- each chunk's
 * record ends come back to the host behind its parse, 8 bytes per record, and are walked while the next
 * chunk is parsed -- so that trailing bytes that are not a record are judged with that window where it
 * really sits, io/buffered.mojo:239-290; -1 (default) = each chunk is judged as a stream of its own;
 * bzq_ingest_next sets it itself);
 * "double_buffer" (1 default / 0: number of output sets, see the lifetime rule at the top);
 * "consumer_guard" (0 default / 1: the library keeps the lifetime rule for work on the consumer stream itself -- every submit records an
 * event on that stream and the ctx stream waits, ON THE DEVICE, for the event of the submit before: a host that enqueues a chunk's
 * consumers on bzq_set_consumer_stream's stream BEFORE it submits the next chunk needs no events of its own and never blocks, and the
 * parse of chunk k+1 still overlaps the consumers of chunk k).
 * Queries (the value is ignored, the answer is the return value): "n_submits", "stream_fallbacks" (chunks parsed twice because pass A's
 * hypothesis failed), "dense_tiles" (tiles of the last parsed chunk that took the serial in-kernel path), "last_folded", "ranks_seen",
 * "device", "numa_node", "numa_cpus", "buf_cache_hits", "buf_cache_held_mb".
 * "state_init_in_kernel" (1 default: the chunk state's initial values are written by the scan's first workgroup; 0: copied in on a side
 * stream, created on first use, as rounds 5-6 shipped it; 2-4: variants of round 6's race hunt), "dump_state" (query,
 * diagnostic: the last result's state snapshots to stderr);
 * environment BZQ_POOL_ZERO=0 restores rounds 4-5's create (the views pool's ticket zeroed on the NULL stream: the defect, kept as the
 * hook of tests/test_gpu_fresh_ctx.py), BZQ_POOL_POISON=1 fills the ticket with garbage first, BZQ_INGEST_RAMP=0 reads a plain file in
 * whole chunks from the first one on (A/B of the short first chunks). */
int32_t bzq_set_option(bzq_ctx* ctx, const char* key, int64_t value);

/* DeviceContext.enqueue_create_host_buffer (record_batch.mojo:316-323): pinned staging the host
 * fills with Reader.read_to_buffer semantics (io/readers.mojo:71-76). */
int32_t bzq_pinned_alloc(size_t bytes, void** out);
int32_t bzq_pinned_free(void* p);

/* Device memory for hosts that have no HIP binding of their own (the plain-C drivers under tests/c_driver; a Mojo host
 * would hand over DeviceContext buffers, record_batch.mojo:364-401): allocation on the ctx's GPU, synchronous copy in. */
int32_t bzq_device_alloc(bzq_ctx* ctx, size_t bytes, void** out);
int32_t bzq_device_free(bzq_ctx* ctx, void* p);
int32_t bzq_copy_to_device(bzq_ctx* ctx, void* d_dst, const void* src, size_t bytes);

/* ---- the hot path --------------------------------------------------------------------------- */

/* Replaces _find_and_consume_ref_record/_scan_record/_validate_fastq_structure (parser.mojo:311-379,
 * utils.mojo:448-551), Validator._validate (record.mojo:162-172), FastqBatch.add
 * (record_batch.mojo:77-87) and the upload (record_batch.mojo:308-411) for a whole chunk.
 * The chunk must start at a record start.  `stream_pos` is the stream offset of data[0] (used for
 * error text only).  is_eof != 0 marks the last chunk of the stream.  Asynchronous on the ctx
 * stream; bzq_chunk_result waits.
 *   _host:   `data` is host memory (pinned for full speed); copied H2D first.
 *   _device: `data` is device memory on the ctx's GPU and is read in place. */
int32_t bzq_submit_chunk_host(bzq_ctx* ctx, const uint8_t* data, uint64_t n, uint64_t stream_pos,
                              int32_t is_eof);
int32_t bzq_submit_chunk_device(bzq_ctx* ctx, const uint8_t* d_data, uint64_t n, uint64_t stream_pos,
                                int32_t is_eof);
/* Blocks until the chunk is parsed; fills `out`.  Returns out->status when it is an error code
 * (>0, not EOF), else 0. */
int32_t bzq_chunk_result(bzq_ctx* ctx, bzq_chunk* out);

/* next_batch (parser.mojo:239-251) + to_device (record_batch.mojo:89-90): records
 * [first_record, first_record+max_records) of the current chunk.  When first_record is a multiple
 * of config.batch_size and max_records <= batch_size (the batches() iteration) the view is zero
 * copy; any other range gets its rebased `ends` / `id_ends` from a small kernel into storage of its
 * own inside the chunk's output set: every view handed out stays valid exactly as long as the
 * chunk's columns, however many views are taken in between. */
int32_t bzq_batch_view(bzq_ctx* ctx, uint64_t first_record, uint32_t max_records, bzq_device_batch* out);
/* batches() (parser.mojo:267-274, _FastqParserBatchIter 700-735) over the current chunk in one call: out[k] = the k-th batch of
 * `max_records` records (the last one shorter), exactly what successive bzq_batch_view(k * max_records, max_records) calls
 * return.  cap = entries available in `out`; *n_out = batches the chunk holds (may exceed cap: only cap are written). */
int32_t bzq_batches(bzq_ctx* ctx, uint32_t max_records, bzq_device_batch* out, uint64_t cap, uint64_t* n_out);
/* The current chunk's chunk-cumulative `ends` / `id_ends` (bzq_chunk.d_ends / d_id_ends): derived from the per-batch arrays by one
 * small kernel the first time it is asked for (the emit kernel writes FastqBatch._ends / _id_ends per batch directly,
 * record_batch.mojo:77-87), free afterwards.  Fills the two pointers of `inout` (may be NULL: then only the ctx's copy of the
 * chunk is updated).  Valid as long as the chunk's other arrays. */
int32_t bzq_chunk_cumulative_ends(bzq_ctx* ctx, bzq_chunk* inout);
/* DeviceFastqBatch.copy_to_host (record_batch.mojo:222-244) */
int32_t bzq_batch_to_host(bzq_ctx* ctx, const bzq_device_batch* batch, bzq_host_batch* out);
/* Copy any device range of the current chunk's arrays to the host (tests, error snippets). */
int32_t bzq_copy_to_host(bzq_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ParseError / ValidationError text of the current chunk's terminal status (errors.mojo:178-234,
 * parser.mojo:276-309, 332-338, 597-610): exactly String(e) of the reference's raise.
 * `records_before` / `lines_before` are the counts delivered by earlier chunks of the same stream.
 * Returns the text length (may exceed cap; output is truncated, always NUL terminated). */
int64_t bzq_format_error(bzq_ctx* ctx, uint64_t records_before, char* buf, size_t cap);

/* ---- multi-GPU shard stitch (new) --------------------------------------------- */

/* Summary of a raw byte shard that does NOT start at a record start.  Filled by
 * bzq_shard_scan; exchanged between ranks by the host (all-gather over RCCL). */
typedef struct bzq_shard_summary {
    uint64_t n_bytes;
    uint64_t n_newlines;
    int64_t first_nl[4];   /* offsets of the first four newlines, -1 if absent */
    uint8_t first_byte, last_byte;
    uint8_t _pad[6];
} bzq_shard_summary;

/* Count newlines of a device-resident shard (one read pass; its tile aggregates are reused by
 * the following bzq_submit_shard). */
int32_t bzq_shard_scan(bzq_ctx* ctx, const uint8_t* d_data, uint64_t n, bzq_shard_summary* out);
/* Parse shard bytes [0, n + halo_bytes) where `lines_before` is the global line index of the
 * shard's first (possibly partial) line and prev_last_byte the byte preceding the shard (0x0A for
 * rank 0).  The halo (the straddling record's remainder received from the next rank) must already
 * sit at d_data + n.  Records whose header starts in [0, n) are delivered. */
int32_t bzq_submit_shard(bzq_ctx* ctx, const uint8_t* d_data, uint64_t n, uint64_t halo_bytes,
                         uint64_t lines_before, uint8_t prev_last_byte, uint64_t stream_pos,
                         int32_t is_last_shard);
/* Number of leading bytes of this shard that belong to the previous rank's last record. */
int32_t bzq_shard_head_bytes(const bzq_shard_summary* s, uint64_t lines_before, uint8_t prev_last_byte,
                             uint64_t* head_bytes);

/* ---- the whole multi-GPU protocol behind the C ABI (SURVEY.md 8b last row, 8e) ------------------
 * One process per GPU; the stream is cut into contiguous BYTE ranges, rank r holds range r in device memory.  A host in
 * any language drives it like the reference drives libz (blazeseq/io/readers.mojo:226-280): no Python, no torch. */

/* ncclUniqueId (rccl.h): created by one rank, handed to the others by the host's own means (a file, MPI, a store). */
typedef struct bzq_nccl_id { char internal[128]; } bzq_nccl_id;
int32_t bzq_comm_get_unique_id(bzq_nccl_id* id_out);
/* RCCL communicator of the ctx (ncclCommInitRank on the ctx's device; librccl.so.1 is bound at this call).  Collectives
 * run on the ctx stream.  One GPU per rank (RCCL refuses two ranks on one device). */
int32_t bzq_comm_init(bzq_ctx* ctx, int32_t rank, int32_t nranks, const void* nccl_id);
/* The same protocol through a POSIX shared-memory segment on the host ("/bzq_<name>"; SURVEY.md 8e fallback via host):
 * same-node ranks under any GPU assignment, several ranks per GPU included.  halo_capacity: largest head a rank may send
 * (0 = 4 MiB); all ranks must pass the same value.  `name` must be unique per job (rank 0 creates the segment and removes
 * it at bzq_comm_destroy; a segment left behind by a crashed job of the same name is replaced). */
int32_t bzq_comm_init_shm(bzq_ctx* ctx, int32_t rank, int32_t nranks, const char* name, uint64_t halo_capacity);
int32_t bzq_comm_destroy(bzq_ctx* ctx);
/* One ring exchange over the communicator with the data checked on arrival (rank r -> r+1; one rank: to itself) and an
 * all-gather of the verdicts: call it once after bzq_comm_init* so that a transport that does not work is reported before the
 * first step, not inside it.  Collective.  0 = every rank received what its neighbour sent. */
int32_t bzq_comm_selftest(bzq_ctx* ctx);

/* What a rank skips, sends and receives; a pure function of the gathered summaries (bzq_plan_shards). */
typedef struct bzq_shard_plan {
    uint64_t lines_before;    /* global line index of the shard's first (possibly partial) line */
    uint64_t head_bytes;      /* leading bytes that belong to a record an earlier rank owns (== n_bytes: the whole shard) */
    uint64_t halo_bytes;      /* bytes received behind the own bytes: the heads of the following ranks, in rank order */
    uint64_t halo_offset;     /* where this rank's head lands in its owner's halo */
    int32_t head_dst;         /* the owner our head goes to, -1 = none */
    int32_t halo_first_src;   /* first rank we receive from (-1 = none) ... */
    int32_t halo_n_src;       /* ... and how many consecutive ranks may contribute (those with head_dst == this rank) */
    uint8_t prev_last_byte;   /* byte preceding the shard in the stream (0x0A for the stream's first byte) */
    uint8_t is_last;          /* no record starts behind this rank's bytes: its parse sees the end of the stream */
    uint8_t _pad[2];
} bzq_shard_plan;
int32_t bzq_plan_shards(const bzq_shard_summary* all, int32_t nranks, bzq_shard_plan* out);

typedef struct bzq_shard_result {
    bzq_chunk chunk;              /* this rank's records (index 0 = its first OWNED record); chunk.status is the rank's own */
    bzq_shard_plan plan;
    uint64_t stream_pos;          /* stream offset of the shard's first byte */
    uint64_t records_before;      /* global index of this rank's record 0 */
    uint64_t global_records, global_bases, global_bytes;   /* sums over all ranks */
    int64_t first_error_record;   /* global index of the stream's first failing record, -1 = none */
    int32_t stream_status;        /* BZQ_EOF, or the FastxErrorCode the sequential parser raises at first_error_record */
    int32_t error_rank;           /* rank that holds the failing record, -1 = none */
} bzq_shard_result;

/* scan -> summary all-gather -> plan -> heads to their owners -> parse [own + halo] -> outcome all-gather, and, only when
 * the stream ends in bytes that are not a record, the reference's BufferedReader window walked through all ranks' record
 * ends in rank order (parser.mojo:464-510, buffered.mojo:276-279) so that BUFFER_EXCEEDED / UNEXPECTED_EOF / an accepted
 * last record come out exactly where the sequential parser gives them.  d_shard: `capacity` >= n + halo bytes of device
 * memory, 16-byte aligned, the rank's n bytes at its start.  Collective: every rank calls it once per step.  Without a
 * communicator (or nranks == 1) the same code runs with no exchange. */
int32_t bzq_shard_stitch(bzq_ctx* ctx, uint8_t* d_shard, uint64_t n, uint64_t capacity, bzq_shard_result* out);
/* File-chunk sharding: bytes [lo, hi) of the file at `path` -- rank r of N takes [size r / N, size (r + 1) / N), aligned to
 * nothing -- into device memory of the ctx, ready for bzq_shard_stitch: n_threads reader threads (0 = 8) pread() 16 MiB pieces
 * into pinned buffers of their own and copy them to the device on streams of their own (what FileReader.read_to_buffer,
 * io/readers.mojo:86-137, does for the reference, for the whole range and with the copy under the next read).  *d_shard:
 * 16-byte aligned, *capacity = the range + max(halo_room, 4 MiB) bytes of room for the halo; owned by the ctx, valid until
 * the next bzq_shard_read_range on it or bzq_destroy.  Not collective: every rank reads for itself. */
int32_t bzq_shard_read_range(bzq_ctx* ctx, const char* path, uint64_t lo, uint64_t hi, uint64_t halo_room, int32_t n_threads,
                             uint8_t** d_shard, uint64_t* n, uint64_t* capacity);
/* records, bases (sequence bytes), bytes over all ranks after the last bzq_shard_stitch */
int32_t bzq_global_counts(bzq_ctx* ctx, uint64_t out[3]);

/* ---- BGZF inflate on the device ("then gzip ingest") ----------------------- */

/* Replaces, for blocked gzip (bgzip / BGZF, SAM spec 4.1), the decompression in front of the parser: RapidgzipReader / GZFile
 * (blazeseq/io/readers.mojo:283-443).  Every block is an independent raw-DEFLATE stream of at most 64 KiB of output; one
 * wave64 decodes one block (blazeseq_amd/csrc/bzq_inflate.hpp), so the COMPRESSED bytes are what crosses PCIe. */
typedef struct bzq_bgzf_block {
    uint64_t comp_offset;   /* of the block's gzip header in the compressed buffer */
    uint32_t comp_size;     /* BSIZE + 1: header (18) + deflate payload + CRC32 + ISIZE */
    uint32_t out_size;      /* ISIZE */
    uint32_t crc32;         /* CRC-32 of the output (the block's trailer, RFC 1952 2.3.1) */
    uint32_t _pad;
    uint64_t out_offset;    /* where its output goes */
} bzq_bgzf_block;

/* Walk the block headers of a HOST buffer: fills blocks[0..*n_blocks) for the whole blocks that fit `n` bytes, `cap` entries
 * and max_out output bytes; *consumed = compressed bytes they span, *out_bytes = their total output.  BZQ_ERR_IO when the
 * bytes at an expected block start are not a BGZF header.  Pure host function. */
int32_t bzq_bgzf_scan(const uint8_t* comp, uint64_t n, uint64_t max_out, bzq_bgzf_block* blocks, int64_t cap, int64_t* n_blocks,
                      uint64_t* consumed, uint64_t* out_bytes);
/* Inflate blocks[0..n_blocks) of the device-resident compressed bytes d_comp[0, comp_bytes) (8 readable bytes of padding
 * behind them are not required) into d_out.  ISIZE, every match distance and length and the CRC-32 of every block's output are checked.
 * Synchronous on the ctx stream.  BZQ_ERR_IO: a block does not decode (bzq_last_error names the first).
 * Precondition (BZQ_ERR_ARG otherwise): the blocks are given in the order of their output -- out_offset must not decrease and
 * the output ranges must not overlap (every block is written by a wave of its own without looking at the others); what
 * bzq_bgzf_scan produces satisfies it. */
int32_t bzq_bgzf_inflate(bzq_ctx* ctx, const uint8_t* d_comp, uint64_t comp_bytes, const bzq_bgzf_block* blocks, int64_t n_blocks,
                         uint8_t* d_out, uint64_t out_capacity);

/* ---- any gzip stream inflated on the device, in parallel (second half) ---------------------------- */

/* Replaces RapidgzipReader -- `RapidgzipFile.open(path, parallelism)` + read_to_buffer, blazeseq/io/readers.mojo:380-443 -- and
 * GZFile (readers.mojo:283-377, libz gzopen / gzread) for gzip files that are not BGZF: single- or multi-member, any
 * compressor, any level.  rapidgzip's two-stage speculation with both stages on the device (blazeseq_amd/csrc/bzq_gzip.hpp):
 * the compressed piece is cut into chunks, one wave per chunk finds a block start and decodes to 16-bit symbols with window
 * markers, the chain of chunks is followed from the piece's exact start, markers are resolved against the 32 KiB in front of
 * every chunk, CRC-32 and ISIZE of every member are checked.  The bytes delivered are the sequential decode's (zlib's).
 * A handle is a stream decoder like zlib's inflate: feed it the file piece by piece, in order. */
typedef struct bzq_gzip bzq_gzip;

typedef struct bzq_gzip_stats {
    uint64_t pieces;             /* bzq_gzip_decode calls that decoded something */
    uint64_t bytes_in;           /* compressed bytes handed in */
    uint64_t bytes_consumed;     /* compressed bytes decoded (the rest waits inside the handle) */
    uint64_t bytes_out;          /* bytes delivered */
    uint64_t chunks;             /* chunks looked at by the block finder */
    uint64_t chunks_with_start;  /* ... in which it found a start */
    uint64_t chain_jobs;         /* decoder runs that ended up in the output (the others were speculation that did not hold) */
    uint64_t fallback_jobs;      /* decoder runs restarted from an explicit position (the speculation had lost the thread) */
    uint64_t members;            /* gzip members completed and verified */
    uint64_t pool_retries;       /* pieces repeated with a larger symbol pool */
} bzq_gzip_stats;

/* A decoder on ctx's device, with a stream of its own. */
int32_t bzq_gzip_open(bzq_ctx* ctx, bzq_gzip** out);
/* "chunk_bytes": compressed bytes per decoder wave (default 16384; 4096 .. 1 MiB).  "host_continuation" (default 1): a stretch of
 * the stream in which the block finder finds nothing to start from -- fixed-Huffman or stored blocks only, or one block of hundreds
 * of KiB -- would be ONE wave's work on the device (~10 MB/s); the decoder stops there (*more = 1) and the next bzq_gzip_decode
 * continues with zlib on the calling thread (200-600 MB/s; same checks, same bytes), for "host_budget_kib" of output (default
 * 32 MiB, doubling while the device keeps handing over), then the device is asked again.  "far_kib" (default 256): how far a
 * decoder goes without meeting a found start before it stops.  Query "host_calls": calls that ran on the host.
 * "predecode" (default 1): with pieces staged ahead (bzq_gzip_stage; up to three buffers: the piece being decoded and two behind
 * it), the finder and the decoders of piece k + 1 are launched as soon as piece k's chain is walked and run beside piece k's last
 * kernels, into a second set of pool / result buffers.  The file pipeline's three (each default 0 on a handle of its own; bzq_ingest_open sets
 * all three on the decoder it owns): "early_find": a staged piece's finder runs behind its copy, on a chunk grid over its own bytes,
 * shifted once the bytes carried over from the piece in front are known.  "chain_l2": beside a predecode, piece k's chain kernels
 * run in the form that fits what the next piece's decoders leave of a CU (256 threads, no LDS, <= 32 VGPRs, windows through the
 * L2) instead of waiting for the decoders to drain.  "defer_verify": a call that launched the next piece's decoders returns
 * WITHOUT waiting for its own last kernels -- d_out is complete in stream order on the handle's stream, not at return -- and the
 * piece's member checks (CRC-32, ISIZE) are made at the start of the next call, which fails if they fail: only for a caller whose
 * consumers of d_out are enqueued on that stream (like gzread, bytes of a damaged member may then have been handed on before the
 * error is).  Query "deferred_calls". */
int32_t bzq_gzip_set_option(bzq_gzip* h, const char* key, int64_t value);
/* The next n compressed bytes (HOST memory; pinned memory makes the copy a DMA) -> their bytes at d_out (DEVICE memory,
 * out_capacity bytes).  Whole DEFLATE blocks only: what is left of the piece stays inside the handle and is decoded in front of
 * the next one.  *more = 1: out_capacity cut the output short, or the next stretch goes to the host (above) -- call again (n = 0
 * is fine) for the rest.  is_last = 1 with the
 * file's last bytes: a stream that does not end there is an error; bytes behind the last member are ignored (as gzread does).
 * Synchronous: on return the *out_bytes bytes are in d_out.  BZQ_ERR_IO: not a valid gzip stream / CRC-32 or length mismatch
 * (bzq_gzip_last_error says which); never wrong bytes. */
int32_t bzq_gzip_decode(bzq_gzip* h, const uint8_t* comp, uint64_t n, int32_t is_last, uint8_t* d_out, uint64_t out_capacity,
                        uint64_t* out_bytes, int32_t* more);
/* Optional read-ahead: a piece that a LATER bzq_gzip_decode will be given starts its way to the device now; the
 * bzq_gzip_decode that gets the same (comp, n) finds it there instead of copying -- and, when it is the piece right behind the
 * one being decoded, with its block finder already run (that starts as soon as the decode in front knows what it leaves over).  comp: pinned host memory, untouched until
 * that call has returned.  Up to three pieces can be outstanding (the one being decoded counts); with all taken the call does
 * nothing.  Pieces are taken in the order staged.  May be called from a second thread while bzq_gzip_decode runs.  A negative
 * return only says that nothing was staged (the piece is then copied by its bzq_gzip_decode); bzq_gzip_last_error is not set. */
int32_t bzq_gzip_stage(bzq_gzip* h, const uint8_t* comp, uint64_t n);
int32_t bzq_gzip_finished(const bzq_gzip* h);   /* 1 once the stream's end has been seen */
int32_t bzq_gzip_get_stats(const bzq_gzip* h, bzq_gzip_stats* out);
const char* bzq_gzip_last_error(const bzq_gzip* h);
void bzq_gzip_close(bzq_gzip* h);

/* ---- host ingest pipeline ---------------------------------------------- */

/* Replaces FileReader.read_to_buffer + BufferedReader._fill_buffer/_compact_from for plain files
 * (blazeseq/io/readers.mojo:86-137, blazeseq/io/buffered.mojo:239-290): a producer thread reads chunk k+1 with
 * n_threads pread() workers into pinned memory while chunk k travels to the device on its own HIP stream and
 * chunk k-1 is consumed; the carry (bytes behind the last record handed out) moves device-to-device. */
typedef struct bzq_ingest bzq_ingest;

typedef struct bzq_ingest_stats {
    uint64_t file_bytes;   /* size of the file */
    uint64_t bytes_read;   /* bytes the reader threads have fetched so far */
    uint64_t chunks;       /* chunks parsed */
    uint64_t records;      /* records the caller has taken */
    double read_s;         /* producer: seconds inside pread() */
    double wait_s;         /* consumer: seconds inside bzq_ingest_next (waiting for H2D + kernels) */
    double total_s;        /* open -> most recent bzq_ingest_next */
    int32_t direct_io;     /* 1: the file is read O_DIRECT (option "ingest_direct" and the filesystem allows it) */
    int32_t numa_node;     /* NUMA node of the GPU the reader threads are bound to, -1 = unknown / not bound */
} bzq_ingest_stats;

/* chunk_bytes 0 = 256 MiB; n_threads <= 0 = 8.  The ctx must outlive the ingest and is used by it.  Options of the ctx read
 * at open: "ingest_direct" = 1: O_DIRECT reads of whole 4 KiB blocks straight into the pinned buffers (files that are not in
 * the page cache; a filesystem that refuses O_DIRECT is read buffered); "ingest_numa" (default 1): the reader threads run on
 * the CPUs of the GPU's NUMA node.
 * The chunk buffers (three pinned host buffers of chunk_bytes + reserve and their device twins; the gzip decoder's pools) are
 * expensive to pin and unpin (~40 ms each way at 256 MiB chunks), so bzq_ingest_close hands them to a process-wide cache and
 * the next open of the process on the same device takes them from there.  Options "pin_cache_bytes" (default 2 GiB) and
 * "dev_cache_bytes" (default 8 GiB) bound what the cache holds, process-wide; 0 gives everything back to the driver now and
 * turns the cache off (environment BZQ_BUF_CACHE=0: off for the whole process). */
int32_t bzq_ingest_open(bzq_ctx* ctx, const char* path, uint64_t chunk_bytes, int32_t n_threads, bzq_ingest** out);
/* Parse the next chunk.  records_taken: how many records of the PREVIOUS chunk the caller consumed (ignored on the
 * first call); the remaining records and the bytes behind them are carried in front of this chunk.  Returns like
 * bzq_chunk_result: 0 = more input follows, > 0 = the stream's terminal FastxErrorCode (BZQ_EOF = clean end; the
 * chunk may still deliver records before it), < 0 runtime failure.  *stream_pos (optional): file offset of the
 * chunk's first byte.  After the terminal chunk every call returns that code with zero records. */
int32_t bzq_ingest_next(bzq_ingest* g, uint64_t records_taken, bzq_chunk* out, uint64_t* stream_pos);
int32_t bzq_ingest_get_stats(const bzq_ingest* g, bzq_ingest_stats* out);
void bzq_ingest_close(bzq_ingest* g);

/* views mode: the records [first_record, first_record + max_records) of the current chunk as spans (zero copy).
 * Mirrors next_view / views() (parser.mojo:159-170, 253-258) for a whole range at once. */
int32_t bzq_views(bzq_ctx* ctx, uint64_t first_record, uint32_t max_records, bzq_device_views* out);

/* FastqBatch.to_device() for a batch that OUTLIVED its chunk (record_batch.mojo:89-90, upload_batch_to_device
 * 404-411: 10 allocations, 10 copies and 3 synchronize() per batch in the reference): one device allocation, five
 * asynchronous copies, one synchronize.  While the chunk is live use bzq_batch_view (zero copy) instead.  The result
 * is owned by the caller: release it with bzq_release_batch (first_record == UINT64_MAX marks an uploaded batch). */
int32_t bzq_upload_batch(bzq_ctx* ctx, const bzq_host_batch* h, bzq_device_batch* out);
int32_t bzq_release_batch(bzq_ctx* ctx, bzq_device_batch* b);

/* ---- device-side consumers of a DeviceFastqBatch ------------------------ */

/* The nw_gpu example on the device batch (examples/nw_gpu/kernels.mojo:21-89, execution.mojo:96-140): global
 * alignment score (match +1, mismatch -1, gap -1) of every record's sequence against `ref` (host bytes).
 * d_scores: device int32[num_records].  Records or references longer than 256 score 0 (kernels.mojo:48-50). */
int32_t bzq_batch_nw_scores(bzq_ctx* ctx, const bzq_device_batch* b, const uint8_t* ref, int32_t ref_len, int32_t* d_scores);
/* Per-record sum of Phred scores, quality byte - b->quality_offset (FastqRecord.phred_scores, record.mojo:340-346;
 * the v0.1 "quality prefix-sum kernel", CHANGELOG.md:73).  d_sums: device int64[num_records].  Asynchronous on the
 * ctx stream. */
int32_t bzq_batch_quality_sums(bzq_ctx* ctx, const bzq_device_batch* b, int64_t* d_sums);
/* Quality distribution per read position (the per-base quality plot; the v0.1 `quality_distribution` example and quality
 * prefix-sum kernel, CHANGELOG.md:73): counts[p * 128 + v] = number of records of the batch whose quality byte at position p
 * (0-based, p < max_positions) equals v (bytes >= 128 count as 127).  counts: host uint64[max_positions * 128].  From it
 * follow the per-position mean / quantiles of Phred (v - quality_offset) and the read-length distribution
 * (records reaching position p = sum over v).  Synchronous. */
int32_t bzq_batch_quality_by_position(bzq_ctx* ctx, const bzq_device_batch* b, int32_t max_positions, uint64_t* counts);
/* The two consumers above for a PIPELINE -- file -> records -> consumer with no host round trip, the reference's only GPU use
 * (examples/nw_gpu/execution.mojo:100-130: next_batch -> batch.to_device(ctx) -> nw_kernel): everything stays on the device and
 * nothing is synchronised, the kernels are enqueued on the consumer stream (bzq_set_consumer_stream; default: the ctx stream).
 * The caller orders them against the parser by the two-chunk lifetime rule: a batch's columns are valid until the SECOND submit
 * after its chunk's, so a consumer of chunk k has to be through before chunk k + 2 is submitted (an event on the consumer stream).
 *   _dev:  the reference bytes are already on the device (d_ref, ref_len <= 256), d_scores: device int32[num_records];
 *   _acc:  d_counts: device uint64[max_positions * 128], ACCUMULATED into (the caller zeroes it once): the per-cycle quality
 *          distribution of a whole file in one table. */
int32_t bzq_batch_nw_scores_dev(bzq_ctx* ctx, const bzq_device_batch* b, const uint8_t* d_ref, int32_t ref_len, int32_t* d_scores);
int32_t bzq_batch_quality_by_position_acc(bzq_ctx* ctx, const bzq_device_batch* b, int32_t max_positions, uint64_t* d_counts);
/* Waits for everything enqueued so far on the consumer stream (bzq_set_consumer_stream; default: the ctx stream): what a host without a
 * HIP binding of its own calls before it reads the asynchronous consumers' outputs (bzq_copy_to_host does not wait for that stream) and
 * before the second submit after a chunk whose consumers may still be running.  DeviceContext.synchronize() of the example,
 * examples/nw_gpu/execution.mojo:126-130. */
int32_t bzq_consumer_synchronize(bzq_ctx* ctx);
/* 256-bin byte histogram of a device column (base composition of sequence_buffer, quality distribution of
 * qual_buffer; the v0.1 quality_distribution example, CHANGELOG.md:73).  hist: host uint64[256]. */
int32_t bzq_column_histogram(bzq_ctx* ctx, const uint8_t* d_col, uint64_t n, uint64_t* hist);
/* Per-record count of G / C bases (either case) of a device byte column whose records are delimited by inclusive running
 * sums d_ends[n_records] (a bzq_device_batch's sequence_buffer + ends, a bzq_fasta_chunk's d_seq_bytes + d_seq_ends):
 * the GC-content consumer of SURVEY 8f rank 2.  d_counts: device int64[n_records]; col_len = d_ends[n_records-1].
 * Synchronous. */
int32_t bzq_column_gc_counts(bzq_ctx* ctx, const uint8_t* d_col, const int64_t* d_ends, int64_t n_records, int64_t col_len,
                             int64_t* d_counts);

/* ---- synthetic input (measurement only) ----------------------------------------------------- */

/* generate_synthetic_fastq_buffer (blazeseq/utils.mojo:831-917) for fixed-length reads, written
 * straight into device memory: records [first, first+count) of a num_reads-record file.
 * Returns bytes written via *out_bytes.  d_out == NULL only sizes. */
int32_t bzq_generate_synthetic_device(bzq_ctx* ctx, int64_t num_reads, int64_t first, int64_t count,
                                      int32_t read_len, int32_t min_phred, int32_t max_phred,
                                      const char* schema, uint8_t* d_out, uint64_t cap, uint64_t* out_bytes);
/* The same with read lengths min_len + ((31 i + 7) mod (max_len - min_len + 1)) (utils.mojo:753-757): BASELINE
 * config 4, long reads of 200..19800 bases. */
int32_t bzq_generate_synthetic_device_var(bzq_ctx* ctx, int64_t num_reads, int64_t first, int64_t count,
                                          int32_t min_len, int32_t max_len, int32_t min_phred, int32_t max_phred,
                                          const char* schema, uint8_t* d_out, uint64_t cap, uint64_t* out_bytes);

/* ---- FASTA records --------------------------------------------
 * Replaces FastaParser.next_record for a whole chunk at a time (blazeseq/fasta/parser.mojo:122-203 over
 * LineIterator.next_line, blazeseq/io/buffered.mojo:600-638): every line stripped of posix spaces, '>' lines open a
 * record, all other lines concatenated into its sequence.  Output is the FastqBatch layout without the quality column:
 * id and sequence bytes back to back + inclusive running sums.  No CPU fallback. */
typedef struct bzq_fasta bzq_fasta;

enum {
    BZQ_FASTA_NO_HEADER = 1,       /* "FASTA: sequence id line does not start with '>'" (parser.mojo:196-200) */
    BZQ_FASTA_EMPTY_SEQUENCE = 11, /* "FASTA record has empty sequence" (parser.mojo:157-165) */
    BZQ_FASTA_NEED_MORE = 12       /* not the last chunk and no record is closed inside it: resubmit with more bytes */
    /* also used: BZQ_OK (more input expected), BZQ_EOF (clean end), BZQ_ASCII_INVALID, BZQ_BUFFER_EXCEEDED
     * ("Line exceeds buffer capacity of N bytes", buffered.mojo:737-765) */
};

typedef struct bzq_fasta_config {
    int32_t check_ascii;     /* fasta ParserConfig.check_ascii (parser.mojo:24-35) */
    int32_t _pad;
    int64_t line_capacity;   /* LineIterator capacity: a line of this many bytes or more is an error.  0 = the
                              * reference's DEFAULT_CAPACITY, 256 KiB (CONSTS.mojo:26); minimum 32768 */
} bzq_fasta_config;

typedef struct bzq_fasta_chunk {
    int32_t status;          /* BZQ_OK / BZQ_EOF / BZQ_FASTA_NEED_MORE, or the code of the first error */
    int32_t _pad;
    int64_t n_records;       /* records delivered: all of them before the first error */
    uint64_t bytes_consumed; /* carry data[bytes_consumed, n) in front of the next chunk (n at EOF) */
    int64_t lines_consumed;  /* '\n' count in [0, bytes_consumed) */
    int64_t seq_bytes, id_bytes;   /* column bytes of the n_records */
    const uint8_t* d_seq_bytes;    /* device; valid until the next bzq_fasta_parse on this handle */
    const uint8_t* d_id_bytes;
    const int64_t* d_seq_ends;     /* [n_records] inclusive running sums of sequence lengths */
    const int64_t* d_id_ends;
    const int64_t* d_hdr_pos;      /* [n_records] chunk offset of each record's '>' */
    int64_t err_record_number, err_line_number, err_file_position;   /* ParseContext of the error, stream-global */
    double kernel_ms;        /* HIP-event time of the kernels of this chunk */
} bzq_fasta_chunk;

int32_t bzq_fasta_create(int32_t device, const bzq_fasta_config* cfg, bzq_fasta** out);
void bzq_fasta_destroy(bzq_fasta* h);
const char* bzq_fasta_last_error(const bzq_fasta* h);   /* h may be NULL: last create() failure */
/* Parses one chunk that starts at a line start.  `data` may be host memory (copied in) or device memory (used in
 * place).  stream_pos / line_base / record_base: bytes, lines and records before this chunk, for error text.
 * Synchronous.  Returns 0 and fills *out also when the stream has an error (out->status); < 0 = runtime failure. */
int32_t bzq_fasta_parse(bzq_fasta* h, const uint8_t* data, uint64_t n, int32_t is_eof, uint64_t stream_pos,
                        uint64_t line_base, uint64_t record_base, bzq_fasta_chunk* out);
/* Diagnostic for a chunk that stopped on an error: index, within that chunk, of the record that was open when the parser
 * stopped -- the record FastaParser.next_record (blazeseq/fasta/parser.mojo:135-170) was still assembling; -1 when the
 * chunk's very first line failed (nothing open yet).  The byte-range shard protocol needs it to restore the sequential
 * parser's order of events across a cut (bzq_fasta_shard_stitch). */
int64_t bzq_fasta_error_open_record(const bzq_fasta* h);
/* The reference's error text for the last chunk's status (errors.mojo:178-234).  Returns the length. */
int32_t bzq_fasta_format_error(bzq_fasta* h, char* buf, size_t cap);
int32_t bzq_fasta_copy_to_host(bzq_fasta* h, void* dst, const void* d_src, size_t bytes);
/* The host ingest pipeline (bzq_ingest_*: reader threads -> pinned double buffers -> copy stream -> device, gzip / BGZF
 * inflated on the way) in front of the FASTA parser: replaces FileReader / GZFile + BufferedReader + LineIterator refills
 * (io/readers.mojo:86-137, 283-443, io/buffered.mojo:239-290, 600-638).  Every record of a chunk is delivered; the bytes of
 * the open record are carried on the device.  out->status is never BZQ_FASTA_NEED_MORE. */
typedef struct bzq_fasta_ingest bzq_fasta_ingest;
int32_t bzq_fasta_ingest_open(bzq_fasta* h, const char* path, uint64_t chunk_bytes, int32_t n_threads, bzq_fasta_ingest** out);
int32_t bzq_fasta_ingest_next(bzq_fasta_ingest* g, bzq_fasta_chunk* out, uint64_t* stream_pos);
int32_t bzq_fasta_ingest_get_stats(const bzq_fasta_ingest* g, bzq_ingest_stats* out);
void bzq_fasta_ingest_close(bzq_fasta_ingest* g);

/* ---- FASTA over byte-range shards (SURVEY.md 8e applied to the 8f rank-4 parser) ----------------------------------
 * New design, the reference is one sequential FastaParser (blazeseq/fasta/parser.mojo:122-203).  The stream is cut into
 * contiguous BYTE ranges, rank r holds range r.  A record belongs to the rank in whose range its header LINE starts; the
 * bytes in front of a rank's first header line (its "head") are the rest of a record an earlier rank owns and travel to
 * that rank (its "halo") -- as many ranks' worth as the record is long.  The stream's first rank owns from offset 0.
 * Every owner then parses [first header line, end of range + halo) as one complete stream: the result over all ranks is
 * the sequential parser's, records in rank order. */
typedef struct bzq_fasta_shard_summary {
    uint64_t n_bytes;
    int64_t first_header;   /* start of the first header line that begins behind a '\n' of this range and shows its '>' in it; -1 */
    int32_t lead_kind;      /* bytes in front of the first '\n' (all, when there is none): 0 first non-space byte is not '>',
                             * 1 it is '>', 2 only spaces and then the '\n', 3 only spaces to the end of the range */
    int32_t last_byte;      /* 10 for an empty range */
    int64_t tail_open;      /* start of the last line when it is non-empty and all spaces so far (a later rank's lead decides
                             * whether it is a header line); -1 otherwise */
} bzq_fasta_shard_summary;

typedef struct bzq_fasta_shard_plan {
    uint64_t stream_pos;       /* stream offset of the range's first byte */
    uint64_t head_bytes;       /* leading bytes that belong to a record of rank head_dst (n_bytes: the rank owns nothing) */
    uint64_t halo_bytes;       /* bytes received behind the own ones */
    uint64_t halo_offset;      /* where this rank's head lands in its owner's halo */
    int32_t head_dst;          /* -1 = no head */
    int32_t halo_first_src, halo_n_src;   /* the halo is the heads of ranks [first, first + n) that name this rank */
    int32_t is_last;           /* the stream's last owner */
} bzq_fasta_shard_plan;

/* Pure function of the gathered summaries (CPU-testable). */
int32_t bzq_fasta_plan_shards(const bzq_fasta_shard_summary* all, int32_t nranks, bzq_fasta_shard_plan* out);
/* Summary of one device-resident range. */
int32_t bzq_fasta_shard_scan(bzq_fasta* h, const uint8_t* d_shard, uint64_t n, bzq_fasta_shard_summary* out);

typedef struct bzq_fasta_shard_result {
    bzq_fasta_chunk chunk;        /* this rank's records; chunk.status is the rank's own (BZQ_EOF = its range parsed clean) */
    bzq_fasta_shard_plan plan;
    uint64_t records_before;      /* global index of this rank's record 0 */
    uint64_t global_records;      /* records the sequential parser delivers before it stops */
    int64_t first_error_record;   /* global index at which the stream fails, -1 = none */
    int32_t stream_status;        /* BZQ_EOF, or the code the sequential parser raises (its text: bzq_fasta_format_error on
                                   * rank error_rank, with stream-global record / line / position numbers) */
    int32_t error_rank;           /* -1 = none */
} bzq_fasta_shard_result;

/* probe -> summary all-gather -> plan -> heads to their owners -> parse -> outcome all-gather, over the communicator of
 * `comm_ctx` (bzq_comm_init / bzq_comm_init_shm; NULL or no communicator = one rank).  d_shard: `capacity` >= n + halo
 * bytes of device memory with the rank's n bytes at its start.  Collective.  A rank behind the stream's first error keeps
 * chunk.n_records = 0. */
int32_t bzq_fasta_shard_stitch(bzq_ctx* comm_ctx, bzq_fasta* h, uint8_t* d_shard, uint64_t n, uint64_t capacity,
                               bzq_fasta_shard_result* out);

/* generate_synthetic_fasta_buffer (blazeseq/utils.mojo:1033-1139), records [first, first+count) of a num_reads-record
 * file, written into device memory.  d_out == NULL only sizes. */
int32_t bzq_fasta_generate_synthetic_device(bzq_fasta* h, int64_t num_reads, int64_t first, int64_t count, int32_t min_len,
                                            int32_t max_len, int32_t line_width, uint8_t* d_out, uint64_t cap,
                                            uint64_t* out_bytes);

#if defined(BZQ_BUILDING) && defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* BLAZESEQ_HIP_H */
